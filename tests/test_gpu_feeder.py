"""Device-side minibatch assembly (eesen_feeder_*, row a1) against the host restatement of
/root/reference/src/netbin/train-ctc-parallel.cc:186-193 (eesen_amd.batching.interleave): bit-exact byte movement."""
import numpy as np
import pytest

from eesen_amd import synth
from eesen_amd.batching import interleave

pytestmark = pytest.mark.gpu


def _mats(rng, lens, D):
    return [rng.standard_normal((t, D)).astype(np.float32) for t in lens]


@pytest.mark.parametrize("lens,D", [([7, 12, 3, 12, 1], 40), ([5], 40), ([9, 4, 9], 13), ([6, 0, 2], 8), ([1000] * 4 + [811, 640], 40),
                                    ([3, 5], 1)])
def test_feeder_matches_host_interleave(gpu, lens, D):
    from eesen_amd.api import Feeder
    rng = np.random.default_rng(5)
    mats = _mats(rng, lens, D)
    want, wl, T = interleave(mats, D)
    f = Feeder()
    slot = f.submit(mats)
    got = f.acquire(slot)
    assert (got.rows, got.cols) == (T * len(lens), D) and got.stride == (D + 3) // 4 * 4
    assert np.array_equal(got.numpy(), want)          # zero padding included
    f.release(slot)


def test_feeder_strided_and_non_float32_inputs(gpu):
    from eesen_amd.api import Feeder
    rng = np.random.default_rng(6)
    big = rng.standard_normal((50, 64)).astype(np.float32)
    mats = [big[:20, :40], big[20:50, 8:48], rng.standard_normal((11, 40))]      # row-strided views and a float64 matrix
    want, _, _ = interleave([np.ascontiguousarray(m, np.float32) for m in mats], 40)
    f = Feeder()
    slot = f.submit(mats)
    assert np.array_equal(f.acquire(slot).numpy(), want)


def test_feeder_slots_rotate_and_overlap(gpu):
    """More batches than slots, submitted ahead of their consumers: every batch must arrive intact, in order."""
    from eesen_amd.api import Feeder
    rng = np.random.default_rng(7)
    f = Feeder(slots=2)
    batches = [_mats(rng, rng.integers(1, 60, size=int(rng.integers(1, 9))), 24) for _ in range(7)]
    staged = f.submit(batches[0])
    for i, b in enumerate(batches):
        cur = staged
        got = f.acquire(cur).numpy().copy()
        f.release(cur)
        if i + 1 < len(batches):
            staged = f.submit(batches[i + 1])      # reuses the other slot while `cur` was just released
        assert np.array_equal(got, interleave(b, 24)[0])


def test_training_from_feeder_equals_training_from_host_matrix(gpu):
    from eesen_amd.api import Net, Ctc, Feeder
    cfg = synth.config("small_bi")
    layers = synth.make_model(**cfg); batch = synth.make_batch(**cfg)
    f3 = batch.feats.reshape(batch.T, batch.S, cfg["D"])
    mats = [np.ascontiguousarray(f3[: batch.lens[s], s, :]) for s in range(batch.S)]
    outs = []
    for via_feeder in (False, True):
        net = Net.from_layers(layers); net.SetTrainOptions(1e-3, 0.9); ctc = Ctc()
        net.SetSeqLengths(batch.lens)
        if via_feeder:
            fd = Feeder(); slot = fd.submit(mats)
            out = net.Propagate(fd.acquire(slot)); fd.release(slot)
        else:
            out = net.Propagate(batch.feats)
        diff = ctc.EvalParallel(batch.lens, out, batch.labels)
        net.Backpropagate(diff)
        outs.append((out.numpy(), net.GetParams()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_feeder_errors(gpu):
    from eesen_amd.api import Feeder, EesenError
    f = Feeder()
    with pytest.raises(EesenError):
        f.submit([np.zeros((0, 8), np.float32)])            # every utterance empty
    with pytest.raises(EesenError):
        f.submit([np.zeros((3, 8), np.float32), np.zeros((3, 9), np.float32)])   # ragged feature dimension
    with pytest.raises(EesenError):
        f.acquire(1)                                        # nothing submitted to that slot
    with pytest.raises(EesenError):
        Feeder(slots=0)
