"""-m gpu: BASELINE.json's full-size configuration (cfg2: 4x512 BiLSTM, S=32, T=1000) checked through
size-independent properties — the CPU oracle needs minutes per step at this size — plus an oracle comparison at
full width but reduced length."""
import numpy as np
import pytest

from eesen_amd import synth
from eesen_amd.parallel import shard_batch
from tests.util import rel_err, valid_mask, split_params

pytestmark = pytest.mark.gpu


def _step(net, ctc, batch, in_diff=False):
    from eesen_amd.api import CuMatrix
    net.SetSeqLengths(batch.lens)
    out = net.Propagate(batch.feats)
    diff = ctc.EvalParallel(batch.lens, out, batch.labels)
    idf = CuMatrix(batch.T * batch.S, batch.feats.shape[1]) if in_diff else None
    net.BackpropagateNoUpdate(diff, idf)
    return out, diff, idf


@pytest.fixture(scope="module")
def cfg2_run(gpu):
    from eesen_amd.api import Net, Ctc
    cfg = synth.config("cfg2")
    layers = synth.make_model(**cfg)
    batch = synth.make_batch(**cfg)
    net = Net.from_layers(layers)
    ctc = Ctc()
    out, diff, idf = _step(net, ctc, batch, in_diff=True)
    return dict(cfg=cfg, layers=layers, batch=batch, net=net, ctc=ctc, out=out.numpy(), diff=diff.numpy(), in_diff=idf.numpy(),
                pzx=ctc.pzx.copy(), grads=net.GetGrads())


def test_cfg2_outputs_are_probabilities_and_gradients_finite(cfg2_run):
    r = cfg2_run; b = r["batch"]
    vm = valid_mask(b.lens, b.T, b.S)
    assert np.all(np.isfinite(r["out"])) and np.all(np.isfinite(r["grads"])) and np.all(np.isfinite(r["pzx"]))
    assert np.max(np.abs(r["out"][vm].sum(1) - 1)) < 1e-5
    assert np.all(r["pzx"] < 0) and np.all(r["pzx"] > -1e4)          # feasible alignments, ln p ~ -T ln K scale
    assert np.all(r["diff"][~vm] == 0) and np.all(r["in_diff"][~vm] == 0)


def test_cfg2_diff_rows_sum_to_zero(cfg2_run):
    """diff = y*sum(gamma) - gamma (ctc-loss.cc:160-168) => every row sums to zero: a checksum over all 32000 x 46 entries."""
    r = cfg2_run
    assert np.max(np.abs(r["diff"].sum(1))) < 2e-5
    # and gamma = y*1 - diff is a posterior: non-negative up to round-off, sums to ~1 on valid frames
    b = r["batch"]; vm = valid_mask(b.lens, b.T, b.S)
    gamma = r["out"][vm] - r["diff"][vm]
    assert gamma.min() > -1e-3 and np.max(np.abs(gamma.sum(1) - 1)) < 1e-2


def test_cfg2_is_deterministic(cfg2_run):
    r = cfg2_run
    out, diff, _ = _step(r["net"], r["ctc"], r["batch"])
    assert np.array_equal(out.numpy(), r["out"]) and np.array_equal(diff.numpy(), r["diff"])
    assert np.array_equal(r["net"].GetGrads(), r["grads"])


def test_cfg2_data_parallel_shards_sum_to_the_full_batch(cfg2_run):
    """SURVEY.md section 8(e) on the HIP path: gradients are sums over frames, so the shards of a 4-way interleaved deal
    (each padded to its own T_max) must add up to the gradient of the whole 32-utterance batch."""
    r = cfg2_run
    total = np.zeros_like(r["grads"], dtype=np.float64); lnp = 0.0
    for rank in range(4):
        sh = shard_batch(r["batch"], rank, 4)
        _step(r["net"], r["ctc"], sh)
        total += r["net"].GetGrads(); lnp += float(r["ctc"].pzx.sum())
    assert abs(lnp - float(r["pzx"].sum())) / abs(float(r["pzx"].sum())) < 1e-5
    for (li, nm, a), (_, _, b) in zip(split_params(r["layers"], total), split_params(r["layers"], r["grads"])):
        assert rel_err(a, b) < 1e-4, f"layer {li} {nm}"


def test_cfg2_loss_gradient_by_finite_differences(cfg2_run):
    """Directional derivative of the full-size loss -sum ln p along the gradient direction, by central differences.
    The step is sized so that the loss moves by ~1, far above the fp32 noise of a loss of magnitude 1e5."""
    from eesen_amd.api import Net, Ctc
    r = cfg2_run; b = r["batch"]
    theta = r["net"].GetParams().astype(np.float64)
    g = r["grads"].astype(np.float64)
    gn = float(np.linalg.norm(g))
    d = g / gn
    eps = 1.0 / gn
    net = Net.from_layers(r["layers"]); ctc = Ctc()

    def loss(th):
        net.SetParams(th.astype(np.float32)); net.SetSeqLengths(b.lens)
        ctc.EvalParallel(b.lens, net.Propagate(b.feats), b.labels)
        return -float(ctc.pzx.astype(np.float64).sum())
    fd = (loss(theta + eps * d) - loss(theta - eps * d)) / (2 * eps)
    assert abs(fd - gn) < 0.02 * gn, f"directional derivative {fd} vs |grad| {gn}"


@pytest.mark.parametrize("name,over", [("cfg2", dict(T=20, S=16)), ("cfg4", dict(T=24, S=8, layers=2)), ("cfg2", dict(T=60, S=5, H=320))])
def test_full_width_short_length_against_oracle(gpu, name, over):
    """Full layer widths (512 / 1024 cells, projection layers, the recipes' 320 cells) at lengths the C oracle finishes in seconds."""
    from eesen_amd.api import Net, Ctc
    from oracle import net as onet
    cfg = synth.config(name); cfg.update(over)
    layers = synth.make_model(**cfg)
    batch = synth.make_batch(**cfg)
    net = Net.from_layers(layers); net.SetTrainOptions(1.0, 0.0); ctc = Ctc()
    out, diff, idf = _step(net, ctc, batch, in_diff=True)
    ora = onet.OracleNet(layers, "f32"); ora.set_train_options(1.0, 0.0)
    o = onet.train_step(ora, batch, "f32")
    vm = valid_mask(batch.lens, batch.T, batch.S)
    assert rel_err(out.numpy()[vm], o["net_out"][vm]) < 1e-4
    assert rel_err(ctc.pzx, o["pzx"]) < 1e-4
    assert rel_err(diff.numpy(), o["diff"]) < 1e-4
    assert rel_err(idf.numpy(), o["in_diff"]) < 1e-4
    for (li, nm, a), (_, _, b) in zip(split_params(layers, net.GetGrads()), split_params(layers, ora.fresh_grads_flat())):
        assert rel_err(a, b) < 1e-4, f"layer {li} {nm}"


# ------------------------------------------------------------------------------------------ BASELINE config 5 (6x1024, S = 64, T <= 3000)
def test_cfg5_width_and_batch_against_oracle_on_the_persistent_kernels(gpu):
    """S = 64 utterances at H = 1024 needs 512 wide-tile workgroups: the persistent kernels run as two sequence windows of 32
    (two cooperative launches per layer pass).  Reduced length, full width and batch, against the oracle; the persistent
    kernels must have been selected for every layer, forward and backward."""
    from eesen_amd.api import Net, Ctc
    from oracle import net as onet
    cfg = synth.config("cfg5"); cfg.update(T=12, layers=2)
    layers = synth.make_model(**cfg)
    batch = synth.make_batch(**cfg)
    net = Net.from_layers(layers); net.SetTrainOptions(1.0, 0.0); ctc = Ctc()
    out, diff, idf = _step(net, ctc, batch, in_diff=True)
    info = net.RecurrenceInfo()
    assert info == dict(lstm_layers=2, fwd_persistent=2, bwd_persistent=2), info
    ora = onet.OracleNet(layers, "f32"); ora.set_train_options(1.0, 0.0)
    o = onet.train_step(ora, batch, "f32")
    vm = valid_mask(batch.lens, batch.T, batch.S)
    assert rel_err(out.numpy()[vm], o["net_out"][vm]) < 1e-4
    assert rel_err(ctc.pzx, o["pzx"]) < 1e-4
    assert rel_err(diff.numpy(), o["diff"]) < 1e-4
    assert rel_err(idf.numpy(), o["in_diff"]) < 1e-4
    for (li, nm, a), (_, _, b) in zip(split_params(layers, net.GetGrads()), split_params(layers, ora.fresh_grads_flat())):
        assert rel_err(a, b) < 1e-4, f"layer {li} {nm}"


def test_cfg5_full_length_layer_persistent_equals_per_step_kernels(gpu, monkeypatch):
    """One 1024-cell BiLSTM layer at the FULL cfg5 size (S = 64, T = 3000): the gate-gradient buffer is 6.3 GB, beyond 32-bit
    buffer offsets (the backward kernel re-bases its resource per chunk of steps), and the batch takes two sequence windows.
    The persistent path must reproduce the one-launch-per-step kernels (forward bit for bit, backward to the last bits)."""
    from eesen_amd.api import Net, Ctc, CuMatrix
    cfg = synth.config("cfg5"); cfg.update(layers=1)
    layers = synth.make_model(**cfg)
    batch = synth.make_batch(**cfg)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("EESEN_PERSISTENT", mode)
        net = Net.from_layers(layers); net.SetTrainOptions(1.0, 0.0); ctc = Ctc()
        net.SetSeqLengths(batch.lens)
        out = net.Propagate(batch.feats)
        diff = ctc.EvalParallel(batch.lens, out, batch.labels)
        idf = CuMatrix(batch.T * batch.S, cfg["D"])
        net.BackpropagateNoUpdate(diff, idf)
        info = net.RecurrenceInfo()
        assert info["fwd_persistent"] == info["bwd_persistent"] == (1 if mode == "1" else 0), info
        res[mode] = (out.numpy(), ctc.pzx.copy(), idf.numpy(), net.GetGrads())
        del net, ctc, out, diff, idf
    # (round 6: the wide forward tile runs on two fp16 planes per operand -- fp32-class, another summation order: 2e-6 like the narrow
    # tile's; EESEN_FWD_SPLIT=0 keeps the fp32-input kernel, which is bit-identical to the per-step one)
    # in_diff: at T = 3000 the CTC stage turns last-bit differences of ln y into 1e-3-class differences of gamma (|alpha| ~ 3e3: the
    # reference's own fp32-vs-fp64 floor for in_diff is 1.6-2.5e-2 here); measured 2.3e-3.  Gradient tensors sum them out.
    e = [rel_err(res["1"][k], res["0"][k]) for k in range(4)]
    assert e[0] < 2e-6 and e[1] < 1e-6 and e[2] < 6e-3 and e[3] < 1e-4, e
    vm = valid_mask(batch.lens, batch.T, batch.S)
    assert np.all(np.isfinite(res["1"][3])) and np.all(res["1"][2][~vm] == 0)
