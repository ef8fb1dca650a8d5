"""Dropout variants of BiLstm(Parallel) (/root/reference/src/net/bilstm-parallel-layer.h:46-94, 209-377,
604-879) on the HIP path.  The oracle's restatement is pinned against the reference run with its own masks
(tests/test_oracle_vs_reference.py::test_dropout_restatement_matches_the_reference); here the HIP library and the oracle get
IDENTICAL masks (injected), so parity is the usual 1e-4; the device-side mask generator is checked on its own."""
import numpy as np
import pytest

from eesen_amd import synth
from tests.util import rel_err, split_params, valid_mask

pytestmark = pytest.mark.gpu
TOL = 1e-4

CASES = {
    "fwd_step": [dict(forward=0.3, fw_step=True), dict(forward=0.2, fw_step=True)],
    "fwd_seq": [dict(forward=0.3, fw_seq=True), {}],
    "rnndrop_step": [dict(recurrent=0.25, rec_step=True, rnndrop=True)] * 2,
    "nml_step": [dict(recurrent=0.3, rec_step=True, nml=True), {}],
    "rnndrop_seq": [dict(recurrent=0.25, rec_seq=True, rnndrop=True), dict(recurrent=0.4, rec_seq=True, rnndrop=True)],
    "nml_seq": [{}, dict(recurrent=0.3, rec_seq=True, nml=True)],
    "both": [dict(forward=0.2, fw_step=True, recurrent=0.25, rec_step=True, rnndrop=True),
             dict(forward=0.1, fw_seq=True, recurrent=0.2, rec_seq=True, nml=True)],
    "twiddle_fwd": [dict(forward=0.2, fw_step=True, recurrent=0.25, rec_step=True, rnndrop=True, twiddle=True)] * 2,
    "twiddle_rec": [dict(forward=0.2, fw_step=True, recurrent=0.25, rec_step=True, nml=True, twiddle=True)] * 2,
}


def _norm(d):
    """dropout options with the factors rounded to fp32 (what a model file or the C-ABI carries)."""
    return None if not d else {k: (float(np.float32(v)) if isinstance(v, float) else v) for k, v in d.items()}


def _draw(rng, rows, cols, p, per_column):
    u = rng.random((1 if per_column else rows, cols))
    m = np.where(u - p > 0, 1.0 / (1.0 - p), 0.0).astype(np.float32)
    return np.repeat(m, rows, axis=0) if per_column else m


def _masks_for(rng, opts, T, S, H, coin):
    """(fwd [T*S x 2H] | None, rec [(T+2)*S or S x 2H] | None) the way the reference shapes them (:46-94)."""
    fwd = rec = None
    tw = opts.get("twiddle", False)
    if opts.get("forward", 0) > 0 and (not tw or coin):
        fwd = _draw(rng, T * S, 2 * H, opts["forward"], opts.get("fw_seq", False))
    if (opts.get("rnndrop") or opts.get("nml")) and (not tw or not coin):
        rows = S if opts.get("rec_seq") else (T + 2) * S
        rec = _draw(rng, rows, 2 * H, opts["recurrent"], opts.get("rec_seq", False))
    return fwd, rec


@pytest.mark.parametrize("persistent", ["1", "0"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_dropout_parity_with_injected_masks(gpu, name, persistent, monkeypatch):
    from eesen_amd.api import Net, Ctc, CuMatrix
    from oracle import net as onet
    monkeypatch.setenv("EESEN_PERSISTENT", persistent)
    cfg = synth.config("small_bi"); cfg.update(T=40, S=32, H=32)      # S = 32, H % 8 == 0: the persistent tiles are used
    layers = synth.make_model(**cfg); batch = synth.make_batch(**cfg)
    lstm = [i for i, L in enumerate(layers) if L["type"].startswith("BiLstm")]
    for li, d in zip(lstm, CASES[name]):
        if d:
            layers[li]["dropout"] = d
    ora = onet.OracleNet(layers, "f32"); ora.set_train_options(1.0, 0.0)
    net = Net.from_layers(layers); net.SetTrainOptions(1.0, 0.0); ctc = Ctc()
    assert [_norm(L.get("dropout")) for L in net.layers()] == [_norm(L.get("dropout")) for L in layers]
    rng = np.random.default_rng(3)
    coin = name == "twiddle_fwd"
    T, S, H = batch.T, batch.S, cfg["H"]
    for li in lstm:
        opts = layers[li].get("dropout") or {}
        if not opts:
            continue
        fwd, rec = _masks_for(rng, opts, T, S, H, coin)
        net.SetDropoutMasks(li, fwd=fwd, rec=rec, twiddle_coin=int(coin))
        ora.set_dropout_masks(li, fwd=fwd, rec_fw=None if rec is None else rec[:, :H], rec_bw=None if rec is None else rec[:, H:],
                              twiddle_apply_forward=coin)
    o = onet.train_step(ora, batch, "f32")
    net.SetSeqLengths(batch.lens)
    out = net.Propagate(batch.feats)
    diff = ctc.EvalParallel(batch.lens, out, batch.labels)
    in_diff = CuMatrix(T * S, cfg["D"])
    net.BackpropagateNoUpdate(diff, in_diff)
    grads = net.GetGrads()
    net.Update()
    vm = valid_mask(batch.lens, T, S)
    assert rel_err(out.numpy()[vm], o["net_out"][vm]) < TOL
    assert rel_err(ctc.pzx, o["pzx"]) < TOL
    assert rel_err(diff.numpy(), o["diff"]) < TOL
    assert rel_err(in_diff.numpy(), o["in_diff"]) < TOL
    for (li, nm, g), (_, _, w) in zip(split_params(layers, grads), split_params(layers, ora.fresh_grads_flat().astype(np.float32))):
        assert rel_err(g, w) < TOL, f"layer {li} {nm}"
    assert rel_err(net.GetParams(), ora.get_params()) < TOL
    # and the dropout really did something: the same net in test mode gives a different output
    net2 = Net.from_layers(layers); net2.SetTestMode(); net2.SetSeqLengths(batch.lens)
    assert rel_err(net2.Propagate(batch.feats).numpy()[vm], o["net_out"][vm]) > 1e-3


def test_generated_masks_statistics_and_determinism(gpu):
    from eesen_amd.api import Net
    cfg = synth.config("small_bi"); cfg.update(T=50, S=16, H=64)
    layers = synth.make_model(**cfg); batch = synth.make_batch(**cfg)
    layers[0]["dropout"] = dict(forward=0.3, fw_step=True, recurrent=0.2, rec_step=True, rnndrop=True)
    layers[1]["dropout"] = dict(forward=0.4, fw_seq=True, recurrent=0.5, rec_seq=True, nml=True)
    T, S, H = batch.T, batch.S, cfg["H"]

    def run(seed):
        net = Net.from_layers(layers); net.SetDropoutSeed(seed); net.SetSeqLengths(batch.lens)
        out = net.Propagate(batch.feats).numpy()
        return out, net.GetDropoutMasks(0, T, S), net.GetDropoutMasks(1, T, S), net

    out_a, m0, m1, net = run(11)
    for m, p in ((m0["fwd"], 0.3), (m0["rec"], 0.2), (m1["fwd"], 0.4), (m1["rec"], 0.5)):
        assert set(np.unique(m)) <= {np.float32(0), np.float32(1.0 / (1.0 - p))}       # Heaviside(u - p) / (1 - p), :59-61
    assert abs((m0["fwd"] > 0).mean() - 0.7) < 0.01 and abs((m0["rec"] > 0).mean() - 0.8) < 0.01     # time-step masks: iid elements
    assert m0["mode"] == 2 and m1["mode"] == 1
    # sequence masks: one draw per column, the same for every row (SetRandUniformCol, cpucompute/matrix.cc:952-965)
    assert np.all(m1["fwd"] == m1["fwd"][0]) and np.all(m1["rec"] == m1["rec"][0])
    assert 0.3 < (m1["fwd"][0] > 0).mean() < 0.9 and 0.2 < (m1["rec"][0] > 0).mean() < 0.8
    assert not np.all(m0["fwd"] == m0["fwd"][0])
    # a pure function of the seed and the draw counter: same seed -> same masks and output, next Propagate -> new masks
    out_b, n0, _, _ = run(11)
    assert np.array_equal(out_a, out_b) and np.array_equal(m0["fwd"], n0["fwd"]) and np.array_equal(m0["rec"], n0["rec"])
    out_c, c0, _, _ = run(12)
    assert not np.array_equal(m0["fwd"], c0["fwd"])
    net.Propagate(batch.feats)
    assert not np.array_equal(net.GetDropoutMasks(0, T, S)["fwd"], m0["fwd"])


def test_generated_masks_end_to_end_against_oracle(gpu):
    """The production path (masks drawn on the device) read back and replayed through the oracle."""
    from eesen_amd.api import Net, Ctc, CuMatrix
    from oracle import net as onet
    cfg = synth.config("small_bi"); cfg.update(T=30, S=8, H=16)
    layers = synth.make_model(**cfg); batch = synth.make_batch(**cfg)
    layers[0]["dropout"] = dict(forward=0.25, fw_step=True, recurrent=0.25, rec_step=True, nml=True)
    layers[1]["dropout"] = dict(recurrent=0.3, rec_seq=True, rnndrop=True)
    T, S, H = batch.T, batch.S, cfg["H"]
    net = Net.from_layers(layers); net.SetTrainOptions(1.0, 0.0); net.SetDropoutSeed(5); ctc = Ctc()
    net.SetSeqLengths(batch.lens)
    out = net.Propagate(batch.feats)
    ora = onet.OracleNet(layers, "f32"); ora.set_train_options(1.0, 0.0)
    for li in (0, 1):
        m = net.GetDropoutMasks(li, T, S)
        step = bool(layers[li]["dropout"].get("rec_step"))
        rec = m["rec"] if step else m["rec"][:S]
        ora.set_dropout_masks(li, fwd=m["fwd"], rec_fw=rec[:, :H], rec_bw=rec[:, H:])
    o = onet.train_step(ora, batch, "f32")
    diff = ctc.EvalParallel(batch.lens, out, batch.labels)
    net.Backpropagate(diff)
    vm = valid_mask(batch.lens, T, S)
    assert rel_err(out.numpy()[vm], o["net_out"][vm]) < TOL and rel_err(diff.numpy(), o["diff"]) < TOL
    assert rel_err(net.GetParams(), ora.get_params()) < TOL


def test_masks_cross_the_boundary_at_the_models_own_width(gpu):
    """A cell count the library pads inside (10 -> 12 per direction): the two mask accessors speak the MODEL's columns, [rows x 2 * 10],
    as include/eesen_hip.h documents them -- injected masks of that width give the oracle's result on the unpadded model, the masks
    read back have that width (and the values that were applied), and a buffer of the padded width is refused (a C caller
    sizing by the model's own H used to overflow its heap on get)."""
    from eesen_amd.api import Net, Ctc, EesenError
    from oracle import net as onet
    cfg = synth.config("small_bi"); cfg.update(T=24, S=6, H=10)
    layers = synth.make_model(**cfg); batch = synth.make_batch(**cfg)
    layers[0]["dropout"] = dict(forward=0.25, fw_step=True, recurrent=0.25, rec_step=True, nml=True)
    layers[1]["dropout"] = dict(recurrent=0.3, rec_seq=True, rnndrop=True)
    T, S, H = batch.T, batch.S, cfg["H"]
    rng = np.random.default_rng(9)
    net = Net.from_layers(layers); net.SetTrainOptions(1.0, 0.0); ctc = Ctc()
    ora = onet.OracleNet(layers, "f32"); ora.set_train_options(1.0, 0.0)
    sent = {}
    for li in (0, 1):
        fwd, rec = _masks_for(rng, layers[li]["dropout"], T, S, H, False)
        sent[li] = (fwd, rec)
        net.SetDropoutMasks(li, fwd=fwd, rec=rec)
        ora.set_dropout_masks(li, fwd=fwd, rec_fw=rec[:, :H], rec_bw=rec[:, H:])
    o = onet.train_step(ora, batch, "f32")
    net.SetSeqLengths(batch.lens)
    out = net.Propagate(batch.feats)
    for li in (0, 1):
        m = net.GetDropoutMasks(li, T, S)
        fwd, rec = sent[li]
        assert m["rec"].shape == ((T + 2) * S, 2 * H)
        assert np.array_equal(m["rec"], rec if rec.shape[0] == (T + 2) * S else np.tile(rec, (T + 2, 1)))
        if fwd is not None:
            assert m["fwd"].shape == (T * S, 2 * H) and np.array_equal(m["fwd"], fwd)
    diff = ctc.EvalParallel(batch.lens, out, batch.labels)
    net.Backpropagate(diff)
    vm = valid_mask(batch.lens, T, S)
    assert rel_err(out.numpy()[vm], o["net_out"][vm]) < TOL and rel_err(diff.numpy(), o["diff"]) < TOL
    assert rel_err(net.GetParams(), ora.get_params()) < TOL
    with pytest.raises(EesenError, match="rec_rows x ndir"):
        net.SetDropoutMasks(0, rec=np.ones(((T + 2) * S, 2 * 12), np.float32))


def test_test_mode_model_files_and_errors(gpu, tmp_path):
    from eesen_amd import nnet_io
    from eesen_amd.api import Net, Ctc, EesenError
    cfg = synth.config("small_bi")
    layers = synth.make_model(**cfg); batch = synth.make_batch(**cfg)
    plain = Net.from_layers(layers); plain.SetSeqLengths(batch.lens)
    want = plain.Propagate(batch.feats).numpy()
    layers[0]["dropout"] = dict(forward=0.5, fw_step=True, recurrent=0.5, rec_step=True, rnndrop=True)
    for binary in (False, True):            # the nine tokens through the library's own reader / writer (bilstm-layer.h:331-373, :435-455)
        p = str(tmp_path / f"d{int(binary)}.nnet")
        nnet_io.write_nnet(p, layers, binary=binary)
        net = Net(); net.Read(p)
        assert net.layers()[0]["dropout"] == layers[0]["dropout"]
        q = str(tmp_path / f"e{int(binary)}.nnet")
        net.Write(q, binary)
        assert nnet_io.read_nnet(q)[0]["dropout"] == layers[0]["dropout"]
        net.SetTestMode(); net.SetSeqLengths(batch.lens)
        assert np.array_equal(net.Propagate(batch.feats).numpy(), want)          # test mode: exactly the no-dropout network
        net.SetTrainMode()
        assert not np.array_equal(net.Propagate(batch.feats).numpy(), want)
    # test mode cannot backpropagate through a dropout layer (bilstm-parallel-layer.h:425)
    net.SetTestMode()
    out = net.Propagate(batch.feats)
    diff = Ctc().EvalParallel(batch.lens, out, batch.labels)
    with pytest.raises(EesenError, match="test mode"):
        net.Backpropagate(diff)
    # recurrent dropout needs exactly one of the step / sequence flags (bilstm-layer.h:102-113)
    bad = Net.from_layers(synth.make_model(**cfg))
    bad.SetLayerDropout(0, dict(recurrent=0.3, rnndrop=True))
    bad.SetSeqLengths(batch.lens)
    with pytest.raises(EesenError, match="exactly one"):
        bad.Propagate(batch.feats)
    with pytest.raises(EesenError):
        bad.SetLayerDropout(2, dict(forward=0.3))            # an AffineTransform has no dropout options
    with pytest.raises(EesenError):
        bad.SetLayerDropout(0, dict(forward=1.0))
