"""CPU: independent ties for the CTC oracle (the reference has no test of its own):
brute-force path enumeration, torch.nn.functional.ctc_loss, and a finite-difference gradient check."""
import itertools

import numpy as np
import pytest

from oracle import net as onet


def _softmax(x):
    e = np.exp(x - x.max(1, keepdims=True))
    return e / e.sum(1, keepdims=True)


def _brute_force_lnp(probs, label):
    T, K = probs.shape
    total = 0.0
    for path in itertools.product(range(K), repeat=T):
        col = [k for k, _ in itertools.groupby(path)]
        if [k for k in col if k != 0] == list(label):
            total += np.prod([probs[t, path[t]] for t in range(T)])
    return np.log(total)


@pytest.mark.parametrize("T,K,label", [(4, 3, [1]), (5, 3, [1, 2]), (5, 3, [1, 1]), (6, 3, [2, 1, 2])])
def test_lnp_equals_path_enumeration(T, K, label):
    rng = np.random.default_rng(T * 10 + K)
    p = _softmax(rng.standard_normal((T, K))).astype(np.float64)
    r = onet.ctc_eval_parallel(p, T, 1, [T], np.array(label, np.int32), np.array([0, len(label)], np.int32), "f64")
    assert abs(r["pzx"][0] - _brute_force_lnp(p, label)) < 1e-10


def test_against_torch_ctc_loss():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(11)
    S, T, K = 4, 30, 9
    lens = np.array([22, 25, 30, 30], np.int32)
    labels = [rng.integers(1, K, size=n).astype(np.int32) for n in (3, 5, 7, 2)]
    labels[1][2] = labels[1][1]  # a repeat
    x = rng.standard_normal((T, S, K))
    logits = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    lp = torch.log_softmax(logits, dim=2)
    loss = torch.nn.functional.ctc_loss(lp, torch.tensor(np.concatenate(labels), dtype=torch.long), torch.tensor(lens, dtype=torch.long),
                                        torch.tensor([len(l) for l in labels], dtype=torch.long), blank=0, reduction="none", zero_infinity=False)
    loss.sum().backward()
    probs = _softmax(x.reshape(T * S, K))
    ids = np.concatenate(labels); off = np.concatenate([[0], np.cumsum([len(l) for l in labels])]).astype(np.int32)
    r = onet.ctc_eval_parallel(probs, T, S, lens, ids, off, "f64")
    assert np.allclose(-r["pzx"], loss.detach().numpy(), rtol=1e-9, atol=1e-9)
    g = logits.grad.numpy().reshape(T * S, K)      # d(-ln p)/d(logits), zero on padded frames
    assert np.allclose(r["diff"], g, rtol=1e-7, atol=1e-9)


def test_finite_difference_of_minus_lnp_wrt_logits():
    rng = np.random.default_rng(3)
    S, T, K = 2, 8, 5
    lens = np.array([6, 8], np.int32)
    ids = np.array([1, 3, 2, 2, 4], np.int32); off = np.array([0, 2, 5], np.int32)
    x = rng.standard_normal((T * S, K))
    f = lambda z: -onet.ctc_eval_parallel(_softmax(z), T, S, lens, ids, off, "f64")["pzx"].sum()
    g = onet.ctc_eval_parallel(_softmax(x), T, S, lens, ids, off, "f64")["diff"]
    eps = 1e-6
    for (r, k) in [(0, 0), (3, 2), (7, 4), (9, 1), (12, 3), (15, 0)]:
        xp = x.copy(); xp[r, k] += eps
        xm = x.copy(); xm[r, k] -= eps
        assert abs((f(xp) - f(xm)) / (2 * eps) - g[r, k]) < 1e-6
