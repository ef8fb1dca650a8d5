"""-m gpu: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.

Tolerance: 1e-4 relative (||a-b||_inf / ||b||_inf per tensor), the figure BASELINE.json's north_star
states for fp32 loss and gradients.
"""
import ctypes as C
import os

import numpy as np
import pytest

from eesen_amd import synth, nnet_io
from tests.util import rel_err, valid_mask, split_params, diff_bound

pytestmark = pytest.mark.gpu
TOL = 1e-4


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 0), (0, 1)])
@pytest.mark.parametrize("M,N,K", [(300, 200, 40), (257, 46, 1024), (130, 1024, 46), (128, 128, 16), (96, 64, 5000),
                                   (33, 17, 7)])
def test_gemm(gpu, a_kc, b_kc, M, N, K):
    from eesen_amd.api import CuMatrix
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    dA = CuMatrix.from_numpy(A if a_kc else np.ascontiguousarray(A.T))
    dB = CuMatrix.from_numpy(np.ascontiguousarray(B.T) if b_kc else B)
    dC = CuMatrix.from_numpy(C0)
    dbias = CuMatrix.from_numpy(bias[None, :])
    alpha, beta = 0.75, 0.5
    rc = gpu.eesen_op_gemm(0, None, a_kc, b_kc, M, N, K, alpha, C.c_void_p(dA.ptr), dA.stride, C.c_void_p(dB.ptr), dB.stride,
                           beta, C.c_void_p(dC.ptr), dC.stride, C.c_void_p(dbias.ptr))
    assert rc == 0, gpu.eesen_last_error()
    want = alpha * (A.astype(np.float64) @ B.astype(np.float64)) + beta * C0 + bias[None, :]
    assert rel_err(dC.numpy(), want) < 2e-6 * max(1, K ** 0.5)


# ------------------------------------------------------------------------------------------ CTC alone
def _random_ctc_case(S, T, K, Umax, seed, min_len_frac=0.6):
    rng = np.random.default_rng(seed)
    lens = np.sort(rng.integers(int(min_len_frac * T), T + 1, size=S)).astype(np.int32)
    lens[-1] = T
    logits = rng.standard_normal((T * S, K)).astype(np.float32) * 2
    probs = np.exp(logits - logits.max(1, keepdims=True)); probs /= probs.sum(1, keepdims=True)
    labels = []
    for s in range(S):
        U = int(rng.integers(1, min(Umax, lens[s] // 2) + 1))
        lab = rng.integers(1, K, size=U).astype(np.int32)
        for i in range(1, U):
            if rng.random() < 0.2: lab[i] = lab[i - 1]
        labels.append(lab)
    return lens, probs.astype(np.float32), labels


@pytest.mark.parametrize("S,T,K,Umax", [(3, 12, 7, 4), (8, 60, 46, 6), (4, 50, 31, 25), (5, 150, 46, 60), (3, 300, 46, 120),
                                        (2, 600, 20, 250), (33, 40, 100, 10)])
def test_ctc_vs_oracle(gpu, S, T, K, Umax):
    from eesen_amd.api import CuMatrix, Ctc
    from oracle import net as onet
    lens, probs, labels = _random_ctc_case(S, T, K, Umax, seed=S * 1000 + T)
    ids = np.concatenate(labels); off = np.concatenate([[0], np.cumsum([len(l) for l in labels])]).astype(np.int32)
    want = onet.ctc_eval_parallel(probs, T, S, lens, ids, off, "f32")
    ctc = Ctc()
    dprob = CuMatrix.from_numpy(probs)
    diff = ctc.EvalParallel(lens, dprob, labels).numpy()
    ctc._rows = T * S
    a, b = ctc.alpha_beta()
    # the lattice: exact sentinel pattern, values to fp32 round-off of the same operation order
    for got, ref in ((a, want["alpha"]), (b, want["beta"])):
        assert np.array_equal(got == -1e30, ref == -1e30)
        m = ref != -1e30
        assert np.max(np.abs(got[m] - ref[m]) / np.maximum(1.0, np.abs(ref[m]))) < 2e-6
    assert rel_err(ctc.pzx, want["pzx"]) < 1e-6
    # diff = y*sum(gamma) - gamma with gamma = exp(alpha + beta - ln p - ln y): in fp32 the exponent carries the
    # round-off of |alpha| (ulp(300) = 3e-5), so two correct fp32 evaluations differ by up to ~ulp(|alpha|) relative.
    # The bar is 1e-4 where that floor allows it, else a small multiple of the fp32 oracle's own distance to fp64.
    arb = onet.ctc_eval_parallel(probs, T, S, lens, ids, off, "f64")
    floor = rel_err(want["diff"], arb["diff"])
    assert rel_err(diff, want["diff"]) < max(TOL, 3 * floor)
    assert rel_err(diff, arb["diff"]) < max(TOL, 3 * floor)
    assert np.all(diff[~valid_mask(lens, T, S)] == 0)
    # greedy decode + edit distance
    ne, nr = ctc.ErrorRateMSeq(lens, dprob, labels)
    assert (ne, nr) == onet.ctc_error_rate_mseq(probs, T, S, lens, ids, off)
    st = ctc.stats()
    assert st["sequences"] == S and st["frames"] == int(lens.sum()) and st["ref_tokens"] == nr


# ------------------------------------------------------------------------------------------ full step
def _run_both(cfg_name, lr, mmt, max_grad, steps=1, seed=777, **over):
    from eesen_amd.api import Net, Ctc, CuMatrix, train_step
    from oracle import net as onet
    cfg = synth.config(cfg_name); cfg.update(over)
    layers = synth.make_model(max_grad=max_grad, seed=seed, **cfg)
    batch = synth.make_batch(seed=seed, **cfg)
    ora = onet.OracleNet(layers, "f32")
    ora.set_train_options(lr, mmt)
    net = Net.from_layers(layers)
    net.SetTrainOptions(lr, mmt)
    ctc = Ctc()
    res = []
    for _ in range(steps):
        o = onet.train_step(ora, batch, "f32")
        net.SetSeqLengths(batch.lens)
        out = net.Propagate(batch.feats)
        diff = ctc.EvalParallel(batch.lens, out, batch.labels)
        in_diff = CuMatrix(batch.T * batch.S, cfg["D"])
        net.BackpropagateNoUpdate(diff, in_diff)
        grads = net.GetGrads()
        net.Update()
        res.append(dict(o=o, net_out=out.numpy(), pzx=ctc.pzx.copy(), diff=diff.numpy(), in_diff=in_diff.numpy(), grads=grads,
                        ora_grads=ora.fresh_grads_flat().astype(np.float32), params=net.GetParams(),
                        ora_params=ora.get_params().astype(np.float32)))
    return layers, batch, res


@pytest.mark.parametrize("cfg_name", ["tiny_bi", "small_bi", "small_uni", "cfg1"])
def test_train_step_parity(gpu, cfg_name):
    layers, batch, res = _run_both(cfg_name, lr=1.0, mmt=0.0, max_grad=0.0)
    r = res[0]; o = r["o"]
    vm = valid_mask(batch.lens, batch.T, batch.S)
    assert rel_err(r["net_out"][vm], o["net_out"][vm]) < TOL
    assert rel_err(r["pzx"], o["pzx"]) < TOL
    assert abs(r["pzx"].sum() - o["pzx"].sum()) / abs(o["pzx"].sum()) < TOL
    bound, diff64 = diff_bound(o["net_out"], batch, o["diff"], TOL)     # 1e-4 unless fp32 itself cannot hold it (T = 200 here)
    assert rel_err(r["diff"], o["diff"]) < bound and rel_err(r["diff"], diff64) < bound
    assert rel_err(r["in_diff"], o["in_diff"]) < bound
    worst = 0.0
    for (li, name, g), (_, _, w) in zip(split_params(layers, r["grads"]), split_params(layers, r["ora_grads"])):
        e = rel_err(g, w); worst = max(worst, e)
        assert e < TOL, f"layer {li} {name}: gradient rel err {e:.2e}"
    # lr = 1, momentum 0, no clipping: theta_after = theta_before - grad
    for (li, name, p), (_, _, w) in zip(split_params(layers, r["params"]), split_params(layers, r["ora_params"])):
        assert rel_err(p, w) < TOL, f"layer {li} {name}: parameter mismatch after update"


def test_momentum_and_clipping(gpu):
    """Three steps with the recipe's settings scaled so that clipping bites (bilstm-layer.h:846-883)."""
    layers, batch, res = _run_both("small_bi", lr=1e-3, mmt=0.9, max_grad=0.05, steps=3)
    for step, r in enumerate(res):
        for (li, name, p), (_, _, w) in zip(split_params(layers, r["params"]), split_params(layers, r["ora_params"])):
            assert rel_err(p, w) < TOL, f"step {step} layer {li} {name}"
    g = res[0]["grads"]
    assert np.max(np.abs(g)) > 0.05, "test must exercise clipping"


def test_unaligned_dims(gpu):
    """Input dim and class count that are not multiples of 4 exercise the padded-row paths."""
    layers, batch, res = _run_both("small_bi", lr=1.0, mmt=0.0, max_grad=0.0, D=13, K=11, S=5, T=21, H=8)
    r = res[0]; o = r["o"]
    vm = valid_mask(batch.lens, batch.T, batch.S)
    assert rel_err(r["net_out"][vm], o["net_out"][vm]) < TOL
    assert rel_err(r["diff"], o["diff"]) < TOL
    assert rel_err(r["in_diff"], o["in_diff"]) < TOL
    assert rel_err(r["grads"], r["ora_grads"]) < TOL


def test_projection_layers(gpu):
    """cfg4's topology at small scale: AffineTransform projections between BiLSTM layers."""
    layers, batch, res = _run_both("small_bi", lr=1.0, mmt=0.0, max_grad=0.0, proj=24, layers=3)
    r = res[0]; o = r["o"]
    assert rel_err(r["pzx"], o["pzx"]) < TOL
    for (li, name, g), (_, _, w) in zip(split_params(layers, r["grads"]), split_params(layers, r["ora_grads"])):
        assert rel_err(g, w) < TOL, f"layer {li} {name}"


def test_sigmoid_and_tanh_layers(gpu, tmp_path):
    """<Sigmoid> / <Tanh> (layer.cc:43-44; sigmoid-layer.h, tanh-layer.h) after the projections: forward, in_diff and every
    gradient against the oracle (itself pinned to the reference's layers in tests/test_oracle_vs_reference.py), and the model
    file with the two parameterless markers through Net::Read / Net::Write."""
    from eesen_amd.api import Net
    layers, batch, res = _run_both("small_bi", lr=1.0, mmt=0.0, max_grad=0.0, proj=24, layers=3, proj_act=["Tanh", "Sigmoid"])
    assert [L["type"] for L in layers].count("Tanh") == 1 and [L["type"] for L in layers].count("Sigmoid") == 1
    r = res[0]; o = r["o"]
    vm = valid_mask(batch.lens, batch.T, batch.S)
    assert rel_err(r["net_out"][vm], o["net_out"][vm]) < TOL
    assert rel_err(r["pzx"], o["pzx"]) < TOL
    assert rel_err(r["in_diff"], o["in_diff"]) < TOL
    for (li, name, g), (_, _, w) in zip(split_params(layers, r["grads"]), split_params(layers, r["ora_grads"])):
        assert rel_err(g, w) < TOL, f"layer {li} {name}"
    p_txt = str(tmp_path / "m.txt"); p_bin = str(tmp_path / "m.bin")
    nnet_io.write_nnet(p_txt, layers, binary=False)
    net = Net().Read(p_txt)
    assert np.array_equal(net.GetParams(), nnet_io.flatten_params(layers))
    net.Write(p_bin, binary=True)
    ours = str(tmp_path / "host.bin")
    nnet_io.write_nnet(ours, layers, binary=True)
    assert open(p_bin, "rb").read() == open(ours, "rb").read()


def test_single_utterance_against_the_reference_nonparallel_path(gpu, tmp_path):
    """S = 1 against what the reference's train-ctc runs for one utterance: the NON-parallel <BiLstm> layers (bilstm-layer.h)
    and Ctc::Eval (ctc-loss.cc:28-75, one-sequence CUDA kernel bodies on the CPU shim) -- not the parallel twins at S = 1.
    Needs oracle/_ref (built in the authoring container, shipped to the GPU box)."""
    from eesen_amd.api import Net, Ctc, CuMatrix
    from oracle import refbind
    if not refbind.available():
        pytest.skip("oracle/_ref not built")
    import tempfile
    cfg = synth.config("small_bi"); cfg.update(S=1, T=45)
    layers = synth.make_model(seed=5, **cfg)
    batch = synth.make_batch(**{**cfg, "seed": 5})
    for L in layers:
        L["type"] = {"BiLstmParallel": "BiLstm"}.get(L["type"], L["type"])
    path = str(tmp_path / "nonpar.nnet")
    nnet_io.write_nnet(path, layers, binary=True)
    ref = refbind.RefNet(path)
    before = ref.get_params()
    ref.set_train_options(1.0, 0.0)
    out_r = ref.propagate(batch.feats)
    c = refbind.cuda_ctc_eval(out_r, batch.labels[0])
    in_r = ref.backpropagate(c["diff"], True)
    grads_r = before.astype(np.float64) - ref.get_params().astype(np.float64)
    net = Net().Read(path)
    net.SetTrainOptions(1.0, 0.0)
    net.SetSeqLengths(batch.lens)
    out = net.Propagate(batch.feats)
    ctc = Ctc()
    diff = ctc.EvalParallel(batch.lens, out, batch.labels)
    in_diff = CuMatrix(batch.T, cfg["D"])
    net.BackpropagateNoUpdate(diff, in_diff)
    assert rel_err(out.numpy(), out_r) < TOL
    assert abs(ctc.pzx[0] - c["pzx"]) < TOL * abs(c["pzx"])
    assert rel_err(diff.numpy(), c["diff"]) < TOL
    assert rel_err(in_diff.numpy(), in_r) < TOL
    # Gradients: every tensor but three.  For the BACKWARD direction's W_m, p_i and p_f the reference's non-parallel layer pairs the
    # gate gradients of time t with the state of time t - 1 (bilstm-layer.h:838,840-841: YM / YC .RowRange(0, T)), although that
    # direction's recurrence source is t + 1 -- its own parallel layer uses the t + 1 rows (bilstm-parallel-layer.h:597-600), and so
    # does this library, which replaces the parallel path (tests/test_oracle_vs_reference.py pins the reference's disagreement
    # with itself; the S = 1 comparison with the PARALLEL reference is test_odd_shapes_and_single_sequence).
    odd = ("Wm_bw", "pi_bw", "pf_bw")
    par = [dict(L, type={"BiLstm": "BiLstmParallel"}.get(L["type"], L["type"])) for L in layers]
    for (li, name, g), (_, _, w) in zip(split_params(par, net.GetGrads()), split_params(par, grads_r.astype(np.float32))):
        if name in odd:
            assert rel_err(g, w) > 1e-2, f"layer {li} {name}: the reference's two layers are expected to disagree here"
        else:
            assert rel_err(g, w) < TOL, f"layer {li} {name}"


def test_nonparallel_markers_are_kept(gpu, tmp_path):
    """A model in the non-parallel form (<BiLstm> / <Lstm>, what format-to-nonparallel writes for decoding) is read like its
    parallel twin (Net::Read maps both onto the same arithmetic, layer.cc:164-170) and written back under its own markers."""
    from eesen_amd.api import Net
    for cfg_name in ("tiny_bi", "small_uni"):
        cfg = synth.config(cfg_name)
        layers = synth.make_model(**cfg)
        for L in layers:
            L["type"] = {"BiLstmParallel": "BiLstm", "LstmParallel": "Lstm"}.get(L["type"], L["type"])
        src, dst = str(tmp_path / f"{cfg_name}.in"), str(tmp_path / f"{cfg_name}.out")
        nnet_io.write_nnet(src, layers, binary=True)
        Net().Read(src).Write(dst, binary=True)
        assert open(src, "rb").read() == open(dst, "rb").read()


def test_model_file_roundtrip(gpu, tmp_path):
    """Net::Read of a text file written by the host tool; Net::Write binary and text; all agree bit for bit."""
    from eesen_amd.api import Net
    cfg = synth.config("tiny_bi")
    layers = synth.make_model(max_grad=50.0, **cfg)
    p_txt = str(tmp_path / "m.txt"); p_bin = str(tmp_path / "m.bin"); p_txt2 = str(tmp_path / "m2.txt")
    nnet_io.write_nnet(p_txt, layers, binary=False)
    net = Net().Read(p_txt)
    assert net.InputDim() == cfg["D"] and net.OutputDim() == cfg["K"]
    assert net.NumParams() == nnet_io.num_params(layers)
    assert np.array_equal(net.GetParams(), nnet_io.flatten_params(layers))
    net.Write(p_bin, binary=True); net.Write(p_txt2, binary=False)
    for p in (p_bin, p_txt2):
        back = nnet_io.read_nnet(p)
        assert [l["type"] for l in back] == [l["type"] for l in layers]
        assert np.array_equal(nnet_io.flatten_params(back), nnet_io.flatten_params(layers))
        assert back[0]["max_grad"] == 50.0
        assert np.array_equal(Net().Read(p).GetParams(), nnet_io.flatten_params(layers))
    # our binary writer and the host tool's writer produce identical bytes
    p_bin2 = str(tmp_path / "m3.bin")
    nnet_io.write_nnet(p_bin2, layers, binary=True)
    assert open(p_bin, "rb").read() == open(p_bin2, "rb").read()


def test_error_behaviour(gpu, tmp_path):
    from eesen_amd.api import Net, Ctc, CuMatrix, EesenError
    cfg = synth.config("tiny_bi")
    layers = synth.make_model(**cfg)
    net = Net.from_layers(layers)
    with pytest.raises(EesenError):       # Propagate before SetSeqLengths
        net.Propagate(np.zeros((6, cfg["D"]), np.float32))
    net.SetSeqLengths([2, 3])
    with pytest.raises(EesenError):       # rows not a multiple of S
        net.Propagate(np.zeros((7, cfg["D"]), np.float32))
    with pytest.raises(EesenError):       # Backpropagate before Propagate
        net.Backpropagate(CuMatrix(6, cfg["K"]))
    bad = str(tmp_path / "bad.txt")
    open(bad, "w").write("<Nnet>\n<BiLstmParallel> <InputDim> 4 <CellDim> 8\n<LearnRateCoef> 1 <MaxGrad> 0 ")
    with pytest.raises(EesenError):       # truncated model file (Net::Read throws, net.cc:279-309)
        Net().Read(bad)
    unk = str(tmp_path / "unknown.txt")
    open(unk, "w").write("<Nnet>\n<Convolutional> <InputDim> 4 <OutputDim> 4\n</Nnet>\n")
    with pytest.raises(EesenError):       # a marker outside the reference's registry (layer.cc:37-46: "Unknown marker") is refused, not skipped
        Net().Read(unk)
    with pytest.raises(EesenError):       # dropout factor outside [0, 1)
        net.SetLayerDropout(0, dict(forward=1.5, fw_step=True))
    with pytest.raises(EesenError):
        Net().Read(str(tmp_path / "does-not-exist"))
    ctc = Ctc()
    with pytest.raises(EesenError):       # label id out of range
        ctc.EvalParallel([3, 3], CuMatrix(6, 5), [[1], [7]])
    with pytest.raises(EesenError):       # empty label sequence (out-of-bounds read in the reference)
        ctc.EvalParallel([3, 3], CuMatrix(6, 5), [[1], []])


@pytest.mark.parametrize("cfg_name,over", [("small_bi", {}), ("small_uni", {}), ("cfg1", {}), ("small_bi", dict(S=37, T=33, H=40)),
                                           ("cfg2", dict(T=50, layers=2)), ("cfg4", dict(T=12, layers=1)),
                                           ("cfg2", dict(T=16, layers=1, H=768)), ("cfg2", dict(T=12, layers=2, H=1024)),
                                           ("cfg2", dict(T=24, layers=2, H=256)), ("cfg2", dict(T=24, layers=1, H=128)),    # 4 x 32 backward tile, 4 / 2 chunks per wave
                                           ("cfg2", dict(T=20, layers=1, H=256, S=12)), ("cfg2", dict(T=20, layers=1, S=30))])  # ragged last tiles / S % 4 != 0
@pytest.mark.parametrize("split", ["0", "1"])
def test_persistent_recurrence_matches_step_kernels(gpu, cfg_name, over, split, monkeypatch):
    """lstm_persistent.hip (one cooperative launch per layer pass, W_m resident in registers, in-kernel hand-off of
    m_t / DG_t) against the one-launch-per-step kernels.  EESEN_FWD_SPLIT=0: the forward recurrence on the fp32-input MFMA, same
    MFMA and reduction order as the per-step kernel, so the forward pass is bit identical; the backward cell equations may be
    FMA-contracted differently by the compiler (last-bit differences).  EESEN_FWD_SPLIT=1 (default): where the narrow tile is taken
    the forward product runs as six bf16 products of exactly split operands (lstm_fwd_persistent_bf_kernel<.., 3, 3>: one fp32
    rounding per product, another summation order) -- equal to the per-step kernels to 2e-6, still bit-identical run to run."""
    from eesen_amd.api import Net, Ctc, CuMatrix
    cfg = synth.config(cfg_name); cfg.update(over)
    layers = synth.make_model(**cfg)
    batch = synth.make_batch(**cfg)
    monkeypatch.setenv("EESEN_FWD_SPLIT", split)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("EESEN_PERSISTENT", mode)
        net = Net.from_layers(layers); net.SetTrainOptions(1e-3, 0.9); ctc = Ctc()
        outs = []
        for it in range(6):  # repeated on unchanged weights: a hand-off race would show up as run-to-run differences
            net.SetSeqLengths(batch.lens)
            out = net.Propagate(batch.feats)
            diff = ctc.EvalParallel(batch.lens, out, batch.labels)
            idf = CuMatrix(batch.T * batch.S, cfg["D"])
            net.BackpropagateNoUpdate(diff, idf)
            outs.append((out.numpy(), diff.numpy(), idf.numpy(), net.GetGrads()))
        net.Update()
        res[mode] = (outs, net.GetParams())
    ref = res["0"][0][0]
    for mode in ("0", "1"):
        for o in res[mode][0]:
            if split == "0":
                assert np.array_equal(o[0], ref[0]) and np.array_equal(o[1], ref[1])   # net_out, diff: bit-identical, every run
            else:
                first = res[mode][0][0]
                assert np.array_equal(o[0], first[0]) and np.array_equal(o[1], first[1])   # bit-identical run to run
                assert rel_err(o[0], ref[0]) < 2e-6 and rel_err(o[1], ref[1]) < 1e-4      # net_out; diff (gamma amplifies last-bit differences of ln y: measured <= 2.2e-5)
            assert np.array_equal(o[2], res[mode][0][0][2]) and np.array_equal(o[3], res[mode][0][0][3])   # deterministic
            tol_b = 5e-6 if split == "0" else 1e-4     # in_diff, gradients (split: they inherit the forward's last-bit differences through the CTC)
            assert rel_err(o[2], ref[2]) < tol_b and rel_err(o[3], ref[3]) < tol_b
    assert rel_err(res["1"][1], res["0"][1]) < (1e-6 if split == "0" else 1e-5)


@pytest.mark.parametrize("arm", [{"EESEN_FWD_SPLIT": "0"}])
def test_forward_recurrence_arms_agree_with_the_default(gpu, arm, monkeypatch):
    """The A/B arm of the narrow forward recurrence that stays in the library -- the fp32-input MFMA tile (EESEN_FWD_SPLIT=0), the kernel
    that is bit-identical to the per-step path -- against the default (three bf16 planes of both operands, six products): the same
    arithmetic up to the order of an fp32 sum, so softmax outputs within 2e-6, gradients within 1e-4, each arm bit-identical run to
    run, every layer pass on a persistent kernel.  (Round 4's other arms -- the 4 x 32 forward tile, two sequence tiles per workgroup
    -- lost their A/B and were removed in round 5; DESIGN.md section 9 keeps the measurements.)"""
    from eesen_amd.api import Net, Ctc
    cfg = synth.config("cfg2"); cfg.update(T=48, layers=2)
    layers = synth.make_model(**cfg)
    batch = synth.make_batch(**cfg)

    def run():
        net = Net.from_layers(layers); net.SetTrainOptions(1e-3, 0.9); ctc = Ctc()
        res = []
        for _ in range(2):
            net.SetSeqLengths(batch.lens)
            out = net.Propagate(batch.feats)
            d = ctc.EvalParallel(batch.lens, out, batch.labels)
            net.BackpropagateNoUpdate(d)
            res.append((out.numpy(), net.GetGrads()))
        ri = net.RecurrenceInfo()
        assert ri["fwd_persistent"] == ri["bwd_persistent"] == ri["lstm_layers"] == 2, ri
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
        return res[0]

    base = run()
    for k, v in arm.items():
        monkeypatch.setenv(k, v)
    got = run()
    assert not np.array_equal(got[0], base[0])          # another kernel really ran
    assert rel_err(got[0], base[0]) < 2e-6 and rel_err(got[1], base[1]) < 1e-4


@pytest.mark.parametrize("over", [dict(S=64, T=48, layers=2), dict(S=64, T=40, layers=2, H=320, D=120), dict(S=48, T=40, layers=1)])
def test_two_narrow_forward_workgroups_per_cu(gpu, over, monkeypatch):
    """--num-sequence 64 on narrow layers (round 5): the bf16-pipe forward tile as ONE grid of up to two workgroups per CU (512 at
    H = 512) where a one-time residency census has seen that many co-resident, instead of the wide 16 x 16 fp32 tile
    (EESEN_FWD_NARROW2=0).  Same arithmetic up to the order of an fp32 sum: softmax outputs within 2e-6, gradients within 1e-4, bit-
    identical run to run, every layer pass persistent, no recovery."""
    from eesen_amd.api import Net, Ctc
    cfg = synth.config("cfg2"); cfg.update(over)
    layers = synth.make_model(**cfg)
    batch = synth.make_batch(**cfg)

    def run():
        net = Net.from_layers(layers); net.SetTrainOptions(1e-3, 0.9); ctc = Ctc()
        res = []
        for _ in range(3):
            net.SetSeqLengths(batch.lens)
            out = net.Propagate(batch.feats)
            d = ctc.EvalParallel(batch.lens, out, batch.labels)
            net.BackpropagateNoUpdate(d)
            res.append((out.numpy(), net.GetGrads()))
        ri = net.RecurrenceInfo()
        assert ri["fwd_persistent"] == ri["bwd_persistent"] == ri["lstm_layers"] == cfg["layers"] and net.recoveries == 0, ri
        for r in res:
            assert np.array_equal(r[0], res[0][0]) and np.array_equal(r[1], res[0][1])
        fwd = net.Plan()["layers"][-1]["forward"]
        return res[0] + ((fwd["kernel"], fwd["workgroups_per_cu"]),)

    base = run()
    monkeypatch.setenv("EESEN_FWD_NARROW2", "0")
    wide = run()
    # another kernel really ran: the narrow tile, two workgroups per CU, against the wide 16 x 16 tile.  (Round 6: both are instantiations
    # of the fp16-plane kernel -- the same products in the same order per output, so the outputs may now agree bit for bit, where
    # rounds 4-5 compared the bf16-plane narrow tile with the fp32-input wide one.)
    assert base[2] != wide[2] and base[2][1] == 2 and wide[2][1] == 1, (base[2], wide[2])
    assert rel_err(base[0], wide[0]) < 2e-6 and rel_err(base[1], wide[1]) < 1e-4


@pytest.mark.parametrize("over", [dict(T=200, S=16, H=64, layers=3), dict(T=130, S=20, H=96, layers=2, min_frac=0.3)])
def test_middle_first_input_gemm_is_bit_identical(gpu, over, monkeypatch):
    """net.cpp "the middle first": the middle rows of the next layer's input GEMM start on the side stream when both chains of the
    running forward recurrence have published step 3T/4 (in-kernel milestone, wait_for_word), the two ends follow.  Shapes where
    the split exists (whole 256-row tiles on both sides of the middle): every run bit-identical to the one-launch GEMM
    (EESEN_FWD_MID=0) -- a GEMM that started before its rows were final would show up here as run-to-run differences."""
    from eesen_amd.api import Net, Ctc
    cfg = synth.config("small_bi"); cfg.update(over)
    assert ((cfg["T"] - 1 - 3 * cfg["T"] // 4) * cfg["S"] + 255) // 256 * 256 < (3 * cfg["T"] // 4 + 1) * cfg["S"] // 256 * 256
    layers = synth.make_model(**cfg)
    batch = synth.make_batch(**cfg)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("EESEN_FWD_MID", mode)
        net = Net.from_layers(layers); ctc = Ctc()
        runs = []
        for it in range(8):
            net.SetSeqLengths(batch.lens)
            out = net.Propagate(batch.feats)
            runs.append(out.numpy())
        info = net.RecurrenceInfo()
        assert info["fwd_persistent"] == info["lstm_layers"] == cfg["layers"]
        outs[mode] = runs
    for r in outs["0"] + outs["1"]:
        assert np.array_equal(r, outs["0"][0])


@pytest.mark.parametrize("rule", ["Adagrad", "RMSProp"])
def test_adaptive_update_rules(gpu, rule, tmp_path):
    """--opt-algorithm Adagrad / RMSProp (trainable-layer.h:65-114): three steps against the oracle, then the
    accumulators through a model file (<BiLstmAccus> / <AffineAccus>) and back."""
    from eesen_amd.api import Net, Ctc
    from oracle import net as onet
    cfg = synth.config("small_bi")
    layers = synth.make_model(max_grad=0.05, learn_rate_coef=0.5, **cfg)    # coef must NOT enter the adaptive rules
    batch = synth.make_batch(**cfg)
    net = Net.from_layers(layers); net.SetTrainOptions(0.002, 0.9); net.SetUpdateAlgorithm(rule, 1e-6, 0.9)
    ora = onet.OracleNet(layers, "f32"); ora.set_train_options(0.002, 0.9); ora.set_update_algorithm(rule, 1e-6, 0.9)
    ctc = Ctc()
    for step in range(3):
        net.SetSeqLengths(batch.lens)
        diff = ctc.EvalParallel(batch.lens, net.Propagate(batch.feats), batch.labels)
        net.Backpropagate(diff)
        onet.train_step(ora, batch, "f32")
        assert rel_err(net.GetParams(), ora.get_params()) < TOL, f"step {step}"
        # squares double the relative error, and a normalised step moves noise-level gradient entries by O(lr): looser bar
        assert rel_err(net.GetAccumulators(), ora.get_accu()) < 2e-3, f"step {step}"
    path = str(tmp_path / "ada.nnet")
    net.Write(path, binary=True)
    back = nnet_io.read_nnet(path)
    assert all(("accu" in L) == bool(L["params"]) for L in back)
    net2 = Net().Read(path)
    assert np.array_equal(net2.GetAccumulators(), net.GetAccumulators()) and np.array_equal(net2.GetParams(), net.GetParams())
    with pytest.raises(Exception):
        net.SetUpdateAlgorithm("Adam")


@pytest.mark.parametrize("over", [dict(T=40, layers=2), dict(T=40, layers=2, H=320, D=120), dict(T=30, layers=1, H=192, S=16), dict(T=30, layers=1, H=448),
                                  dict(T=30, layers=2, H=320, D=120, S=10), dict(T=30, layers=1, S=22), dict(T=30, layers=1, S=13, H=256)])
def test_backward_tiles_agree_and_are_deterministic(gpu, over, monkeypatch):
    """The 4-sequence x 32-unit backward tile (default where the shape allows) and the 8-sequence tile it replaced
    (EESEN_BWD_Q4=0: v_mfma_f32_4x4x1 with CBSZ = 2, part of W_m^T in LDS) on the same inputs: each bit-identical run after
    run (the in-kernel hand-off leaves no room for a race), and equal to each other up to the summation order.  Since round 5 also
    at cell counts that are not multiples of 128 (the recipes' 320; 192; 448): K = 4H then does not fill the waves' chunk pairs -- and
    at sequence counts that are not multiples of 4 (the recipes' default --num-sequence 10): the last tile is ragged."""
    from eesen_amd.api import Net, Ctc, CuMatrix
    cfg = synth.config("cfg2"); cfg.update(over)      # S = 32, bidirectional
    layers = synth.make_model(**cfg); batch = synth.make_batch(**cfg)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("EESEN_BWD_Q4", mode)
        net = Net.from_layers(layers); ctc = Ctc()
        runs = []
        for _ in range(4):
            net.SetSeqLengths(batch.lens)
            out = net.Propagate(batch.feats)
            diff = ctc.EvalParallel(batch.lens, out, batch.labels)
            idf = CuMatrix(batch.T * batch.S, cfg["D"])
            net.BackpropagateNoUpdate(diff, idf)
            runs.append((idf.numpy(), net.GetGrads()))
        assert net.RecurrenceInfo()["bwd_persistent"] == cfg["layers"]
        res[mode] = runs
    for mode in ("0", "1"):
        for r in res[mode]:
            assert np.array_equal(r[0], res[mode][0][0]) and np.array_equal(r[1], res[mode][0][1])
    assert not np.array_equal(res["1"][0][1], res["0"][0][1])     # two different kernels ran
    assert rel_err(res["1"][0][0], res["0"][0][0]) < 1e-5 and rel_err(res["1"][0][1], res["0"][0][1]) < 1e-5

@pytest.mark.parametrize("over", [dict(T=24, layers=2, H=1024), dict(T=20, layers=1, H=1024, S=64), dict(T=20, layers=1, H=1024, S=24)])
def test_wide_backward_tile_on_fp16_planes(gpu, over, monkeypatch):
    """lstm_bwd_persistent_ksplit_h_kernel (round 6, EESEN_BWD_F16): the K-split backward tile of 1024-cell layers with W_m^T and the
    gate gradients as two fp16 planes each (three products), the gate gradients published a second time as planes with a power of
    two per wave of their producer.  Against the fp32-input K-split tile (EESEN_BWD_F16=0; at S = 64 its time-multiplexed form) on
    the same inputs: input gradient and every parameter gradient within 2e-5 (fp32-class products, another summation order), each arm
    bit-identical run after run, every layer pass persistent, no recovery; ragged sequence tiles (S = 24) included."""
    from eesen_amd.api import Net, Ctc, CuMatrix
    cfg = synth.config("cfg2"); cfg.update(over)
    layers = synth.make_model(**cfg); batch = synth.make_batch(**cfg)
    res, kern = {}, {}
    for mode in ("0", "1"):
        monkeypatch.setenv("EESEN_BWD_F16", mode)
        net = Net.from_layers(layers); ctc = Ctc()
        runs = []
        for _ in range(3):
            net.SetSeqLengths(batch.lens)
            out = net.Propagate(batch.feats)
            diff = ctc.EvalParallel(batch.lens, out, batch.labels)
            idf = CuMatrix(batch.T * batch.S, cfg["D"])
            net.BackpropagateNoUpdate(diff, idf)
            runs.append((idf.numpy(), net.GetGrads()))
        info = net.RecurrenceInfo()
        assert info["bwd_persistent"] == info["lstm_layers"] == cfg["layers"] and net.recoveries == 0, (mode, info)
        for r in runs:
            assert np.array_equal(r[0], runs[0][0]) and np.array_equal(r[1], runs[0][1]), mode
        res[mode] = runs[0]
        kern[mode] = net.Plan()["layers"][-1]["backward"]["kernel"]
    assert "ksplit_h" in kern["1"] and "ksplit_h" not in kern["0"] and "ksplit" in kern["0"], kern
    assert np.isfinite(res["1"][1]).all() and np.abs(res["1"][1]).max() > 0
    assert rel_err(res["1"][0], res["0"][0]) < 2e-5 and rel_err(res["1"][1], res["0"][1]) < 2e-5


@pytest.mark.parametrize("over", [dict(T=40, layers=2), dict(T=33, layers=1, H=256, S=24), dict(T=36, layers=2, S=64),
                                  dict(T=33, layers=2, H=320, D=120, S=32), dict(T=33, layers=2, H=320, D=120, S=64), dict(T=33, layers=1, S=52)])
def test_two_sequence_tiles_per_backward_workgroup(gpu, over, monkeypatch):
    """lstm_bwd_persistent_q4_kernel<., 8> (round 5): two 4-sequence tiles per workgroup against the same resident W_m^T -- what a
    narrow layer takes at --num-sequence 64, where the one-tile grid needs 512 workgroups.  Per tile the same instructions in the
    same order: forced on shapes where the one-tile form runs too (EESEN_BWD_Q4_ST8=2) the input gradient and every parameter
    gradient must be BIT-identical to it, run after run; at S = 64 (the default there) it is held against the 16 x 16 tile it replaces
    (EESEN_BWD_Q4_ST8=0) up to the summation order."""
    from eesen_amd.api import Net, Ctc, CuMatrix
    cfg = synth.config("cfg2"); cfg.update(over)
    layers = synth.make_model(**cfg); batch = synth.make_batch(**cfg)
    res = {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("EESEN_BWD_Q4_ST8", mode)
        net = Net.from_layers(layers); ctc = Ctc()
        runs = []
        for _ in range(3):
            net.SetSeqLengths(batch.lens)
            out = net.Propagate(batch.feats)
            diff = ctc.EvalParallel(batch.lens, out, batch.labels)
            idf = CuMatrix(batch.T * batch.S, cfg["D"])
            net.BackpropagateNoUpdate(diff, idf)
            runs.append((idf.numpy(), net.GetGrads()))
        info = net.RecurrenceInfo()
        assert info["bwd_persistent"] == info["lstm_layers"] and net.recoveries == 0, (mode, info)
        for r in runs:
            assert np.array_equal(r[0], runs[0][0]) and np.array_equal(r[1], runs[0][1]), mode
        res[mode] = runs[0]
    assert np.isfinite(res["1"][1]).all() and np.abs(res["1"][1]).max() > 0
    if cfg["S"] <= 32:    # default = the one-tile form: the forced two-tile form must reproduce it bit for bit
        assert np.array_equal(res["2"][0], res["1"][0]) and np.array_equal(res["2"][1], res["1"][1])
        assert np.array_equal(res["0"][0], res["1"][0])       # (the switch's value 0 changes nothing where one tile fits)
    else:                 # default = the two-tile form; 0 = the 16 x 16 tile
        assert np.array_equal(res["2"][0], res["1"][0]) and np.array_equal(res["2"][1], res["1"][1])
        assert not np.array_equal(res["0"][1], res["1"][1])   # another kernel really ran
        assert rel_err(res["1"][0], res["0"][0]) < 1e-5 and rel_err(res["1"][1], res["0"][1]) < 1e-5


def test_persistent_kernel_gives_up_loudly_instead_of_hanging(gpu, monkeypatch, capfd):
    """A hand-off that cannot complete (here: a spin bound of zero polls) must surface at the next synchronisation point --
    never as a hang or as silently wrong numbers: a WARNING on stderr and a fall-back to the per-step kernels (an exception
    when a data-parallel communicator is attached, tests/test_gpu_comm.py), after which the handle works."""
    from eesen_amd.api import Net
    cfg = synth.config("small_bi")
    layers = synth.make_model(**cfg); batch = synth.make_batch(**cfg)
    monkeypatch.setenv("EESEN_SPIN_LIMIT", "0")
    net = Net.from_layers(layers)
    monkeypatch.delenv("EESEN_SPIN_LIMIT")
    net.SetSeqLengths(batch.lens)
    net.Propagate(batch.feats)
    net.Synchronize()
    assert "gave up waiting for a peer workgroup" in capfd.readouterr().err
    net.RecurrenceInfo()
    assert net.recoveries == 1
    net.SetSeqLengths(batch.lens)
    out = net.Propagate(batch.feats).numpy()
    assert net.RecurrenceInfo()["fwd_persistent"] == 0
    monkeypatch.setenv("EESEN_PERSISTENT", "0")
    ok = Net.from_layers(layers); ok.SetSeqLengths(batch.lens)
    assert np.array_equal(out, ok.Propagate(batch.feats).numpy())


def test_first_poll_delays_do_not_follow_a_bimodal_reading(gpu, monkeypatch, capfd):
    """The hand-off waits' first-poll delays (csrc/net.cpp): the values tuned on the MI355X -- 400 ns forward, 280 backward, 420 beside
    side-stream GEMMs -- whenever the increment flight the Net measures at creation is in the part's own range (300-700 ns: the reading has
    two modes, 370-420 and 550-620 ns, that the step time does not share; round 6 scaled the delays with it and lost 0.3-0.9 ms per step
    in far-mode processes); EESEN_POLL_NS overrides both backward values; results are the same bits whatever the delays."""
    import re
    from eesen_amd.api import Net
    cfg = synth.config("small_bi")
    layers = synth.make_model(**cfg); batch = synth.make_batch(**cfg)
    monkeypatch.setenv("EESEN_PRINT_FLIGHT", "1")

    def make():
        net = Net.from_layers(layers)
        m = re.search(r"increment flight (\d+) ns; first-poll delays forward (\d+), backward (\d+) \(beside side-stream GEMMs (\d+)\) ns", capfd.readouterr().err)
        assert m, "the Net did not report its delays"
        net.SetSeqLengths(batch.lens)
        return net, tuple(int(g) for g in m.groups())

    net, (flight, fwd, bwd, side) = make()
    assert 100 <= flight <= 2000
    if 300 <= flight <= 700:
        assert (fwd, bwd, side) == (400, 280, 420)
    else:   # another clock / fabric: everything scales with the reading
        assert abs(fwd - flight) <= 10 and abs(bwd - 0.7 * flight) <= 10 and abs(side - 1.05 * flight) <= 10
    base = net.Propagate(batch.feats).numpy()
    monkeypatch.setenv("EESEN_POLL_NS", "50,900")
    net2, (_, fwd2, bwd2, side2) = make()
    assert (fwd2, bwd2, side2) == (50, 900, 900)
    assert np.array_equal(net2.Propagate(batch.feats).numpy(), base)
    assert net2.RecurrenceInfo()["fwd_persistent"] == net.RecurrenceInfo()["fwd_persistent"] > 0


@pytest.mark.parametrize("kind,H,S,T", [("BiLstmParallel", 10, 8, 30), ("LstmParallel", 7, 4, 21), ("BiLstmParallel", 150, 16, 40)])
def test_cell_counts_that_are_not_multiples_of_4(gpu, kind, H, S, T, tmp_path):
    """The kernels fetch the recurrent state four cells at a time; the reference takes any <CellDim>.  The library pads such a layer
    INSIDE (zero cells that stay zero: Layer::din_f / dout_f / Hf in csrc/net.h) and maps at the parameter / model-file boundary, so
    the ORIGINAL model goes in and comes out: against the oracle on the original model through three SGD steps with momentum and
    clipping; GetParams / Write in the file's own dimensions; and bit for bit what `model_tools pad-cells` + the library gives
    (the same internal net), whose trained file unpad-cells cuts back only if every padded entry is still exactly zero."""
    from eesen_amd import model_tools, nnet_io
    from eesen_amd.api import Net, Ctc, CuMatrix
    from oracle import net as onet
    cfg = dict(kind=kind, layers=2, H=H, D=13, K=11, S=S, T=T)
    layers = synth.make_model(max_grad=5.0, **cfg); batch = synth.make_batch(**cfg)
    padded = model_tools.pad_cells_layers(layers)
    ora = onet.OracleNet(layers, "f32"); ora.set_train_options(0.02, 0.9)
    nets = [Net.from_layers(layers), Net.from_layers(padded)]
    assert nets[0].GetParams().size == nnet_io.flatten_params(layers).size
    assert np.array_equal(nets[0].GetParams(), nnet_io.flatten_params(layers))
    for n in nets:
        n.SetTrainOptions(0.02, 0.9)
    ctc = Ctc()
    vm = valid_mask(batch.lens, batch.T, batch.S)
    for step in range(3):
        o = onet.train_step(ora, batch, "f32")
        got = []
        for n in nets:
            n.SetSeqLengths(batch.lens)
            out = n.Propagate(batch.feats)
            diff = ctc.EvalParallel(batch.lens, out, batch.labels)
            in_diff = CuMatrix(batch.T * batch.S, cfg["D"])
            n.Backpropagate(diff, in_diff)
            got.append((out.numpy(), ctc.pzx.copy(), in_diff.numpy()))
        assert all(np.array_equal(a, b) for a, b in zip(*got)), step          # padded inside == padded by the tool
        bar = TOL if step == 0 else 5 * TOL      # (two trajectories in fp32: the later steps carry the earlier steps' rounding)
        assert rel_err(got[0][0][vm], o["net_out"][vm]) < bar, step
        assert rel_err(got[0][1], o["pzx"]) < bar, step
        assert rel_err(got[0][2], o["in_diff"]) < 3 * bar, step
    paths = [str(tmp_path / "direct.nnet"), str(tmp_path / "tool.nnet")]
    for n, pth in zip(nets, paths):
        n.Write(pth, binary=True)
    direct = nnet_io.read_nnet(paths[0])
    assert [(L["type"], L["input_dim"], L["output_dim"]) for L in direct] == [(L["type"], L["input_dim"], L["output_dim"]) for L in layers]
    cut = model_tools.unpad_cells_layers(nnet_io.read_nnet(paths[1]), [H, H])      # raises unless every padded entry is still exactly 0
    for La, Lb, Lc in zip(ora.to_layers(), cut, direct):
        for a, b, c in zip(La["params"], Lb["params"], Lc["params"]):
            assert np.array_equal(b, c), La["type"]
            assert rel_err(b, a) < 5 * TOL, La["type"]
    info = nets[0].Info() if hasattr(nets[0], "Info") else None
    assert info is None or "nan" not in str(info).lower()


@pytest.mark.parametrize("kind,H", [("BiLstmParallel", 6), ("LstmParallel", 10)])
def test_a_padded_lstm_layer_as_the_whole_net(gpu, kind, H):
    """Seam 2 runs ONE LSTM layer per handle (include/eesen_hip_layer.h): with a cell count that is not a multiple of 4 the handle's
    output and out_diff still have the file's columns -- the padded cells are gathered out / scattered in at the net's ends."""
    from eesen_amd.api import Net, CuMatrix
    from oracle import net as onet
    cfg = dict(kind=kind, layers=1, H=H, D=5, K=4, S=3, T=9)
    layers = synth.make_model(**cfg)[:1]; batch = synth.make_batch(**cfg)
    width = layers[0]["output_dim"]
    ora = onet.OracleNet(layers, "f32"); ora.set_train_options(1.0, 0.0); ora.set_seq_lengths(batch.lens)
    want = ora.propagate(batch.feats)
    od = np.random.default_rng(3).standard_normal(want.shape).astype(np.float32)
    od[~valid_mask(batch.lens, batch.T, batch.S)] = 0.0
    want_in = ora.backpropagate(od, update=False)
    net = Net.from_layers(layers); net.SetSeqLengths(batch.lens)
    out = net.Propagate(batch.feats)
    assert out.numpy().shape == (batch.T * batch.S, width)
    vm = valid_mask(batch.lens, batch.T, batch.S)
    assert rel_err(out.numpy()[vm], want[vm]) < TOL
    in_diff = CuMatrix(batch.T * batch.S, cfg["D"])
    net.BackpropagateNoUpdate(CuMatrix.from_numpy(od), in_diff)
    assert rel_err(in_diff.numpy(), want_in) < TOL
    assert rel_err(net.GetGrads(), ora.fresh_grads_flat().astype(np.float32)) < TOL


def test_cell_padding_is_refused_where_it_would_show(gpu):
    from eesen_amd.api import Net
    z = lambda *sh: np.zeros(sh, np.float32)
    lstm = dict(type="BiLstmParallel", input_dim=5, output_dim=12, params=[z(24, 5), z(24, 6), z(24), z(6), z(6), z(6)] * 2)
    with pytest.raises(Exception, match="Sigmoid"):
        Net.from_layers([lstm, dict(type="Sigmoid", input_dim=12, output_dim=12, params=[])])
    with pytest.raises(Exception, match="Softmax"):
        Net.from_layers([lstm, dict(type="Softmax", input_dim=12, output_dim=12, params=[])])
    ok = Net.from_layers([lstm, dict(type="Tanh", input_dim=12, output_dim=12, params=[]),
                          dict(type="AffineTransform", input_dim=12, output_dim=4, params=[z(4, 12), z(4)]),
                          dict(type="Softmax", input_dim=4, output_dim=4, params=[])])
    assert ok.GetParams().size == 2 * (24 * 5 + 24 * 6 + 24 + 18) + 4 * 12 + 4
    assert Net.from_layers([lstm]).GetParams().size == 2 * (24 * 5 + 24 * 6 + 24 + 18)


@pytest.mark.parametrize("over", [dict(S=1, T=37), dict(S=2, T=2), dict(S=17, T=9, H=20), dict(S=33, T=5, H=36, layers=1),
                                  dict(S=6, T=40, min_frac=0.2)])
def test_odd_shapes_and_single_sequence(gpu, over):
    """Single-sequence training (the reference's `train-ctc` is this path at S = 1), partially filled sequence / unit tiles,
    very short T and heavily ragged lengths, against the oracle."""
    cfg = dict(synth.config("small_bi")); cfg.update(over)
    layers, batch, res = _run_both("small_bi", lr=1.0, mmt=0.0, max_grad=0.0, **over)
    r = res[0]; o = r["o"]
    vm = valid_mask(batch.lens, batch.T, batch.S)
    assert rel_err(r["net_out"][vm], o["net_out"][vm]) < TOL
    assert rel_err(r["pzx"], o["pzx"]) < TOL
    assert rel_err(r["diff"], o["diff"]) < TOL
    assert rel_err(r["in_diff"], o["in_diff"]) < TOL
    assert rel_err(r["grads"], r["ora_grads"]) < TOL


@pytest.mark.parametrize("S", [10, 20])
def test_recipe_shape_on_the_persistent_kernels(gpu, S):
    """The reference's OWN recipe shape (asr_egs/wsj/run_ctc_phn.sh:65-85, steps/train_ctc_parallel.sh:13-21: 4 x 320 BiLSTM on
    120-d features, --num-sequence 10 / 20) against the oracle.  H = 320 is a multiple of neither 128 nor 256, S fills neither a
    16- nor a 32-sequence tile: none of the tiles the BASELINE configurations take applies, and the layer passes must still run
    on the persistent kernels (bench.py's recipe leg times exactly these)."""
    from eesen_amd.api import Net, Ctc, CuMatrix
    from oracle import net as onet
    over = dict(layers=4, H=320, D=120, K=46, S=S, T=48)
    cfg = dict(synth.config("small_bi")); cfg.update(over)
    layers = synth.make_model(max_grad=0.0, **cfg)
    batch = synth.make_batch(**cfg)
    ora = onet.OracleNet(layers, "f32"); ora.set_train_options(1.0, 0.0)
    o = onet.train_step(ora, batch, "f32")
    net = Net.from_layers(layers); net.SetTrainOptions(1.0, 0.0)
    ctc = Ctc()
    net.SetSeqLengths(batch.lens)
    out = net.Propagate(batch.feats)
    diff = ctc.EvalParallel(batch.lens, out, batch.labels)
    in_diff = CuMatrix(batch.T * batch.S, cfg["D"])
    net.BackpropagateNoUpdate(diff, in_diff)
    info = net.RecurrenceInfo()
    assert info["fwd_persistent"] == info["bwd_persistent"] == info["lstm_layers"] == 4, info
    vm = valid_mask(batch.lens, batch.T, batch.S)
    assert rel_err(out.numpy()[vm], o["net_out"][vm]) < TOL
    assert rel_err(ctc.pzx, o["pzx"]) < TOL
    assert rel_err(diff.numpy(), o["diff"]) < TOL
    assert rel_err(in_diff.numpy(), o["in_diff"]) < TOL
    for (li, name, g), (_, _, w) in zip(split_params(layers, net.GetGrads()), split_params(layers, ora.fresh_grads_flat().astype(np.float32))):
        assert rel_err(g, w) < TOL, f"layer {li} {name}: gradient rel err {rel_err(g, w):.2e}"


def test_results_do_not_depend_on_the_allocation_history(gpu):
    """The recipes sort their lists by length, so the minibatches of an epoch grow; device buffers grow by half when they have to
    (round 5: exact growth freed and re-allocated every buffer on every minibatch) and stay when a shorter minibatch follows.  What a
    minibatch computes must not depend on what the handle saw before: pad columns of over-allocated matrices are zero, boundary row
    blocks are re-zeroed at the new T, and the GEMMs' split-K factor follows the minibatch's own workspace request, not the
    allocation -- gradients of the SAME minibatch are bit-identical on a fresh handle, after a longer one, and after a run of
    growing ones."""
    from eesen_amd.api import Net, Ctc, CuMatrix
    cfg = synth.config("cfg2"); cfg.update(layers=2, H=64, K=30)     # K = 30: the output matrices have pad columns (ld 32)
    layers = synth.make_model(**cfg)
    shapes = [(16, 30), (20, 44), (32, 60), (32, 90), (24, 140)]     # (S, T): growing, as in a length-sorted epoch
    batches = [synth.make_batch(**{**cfg, "S": S, "T": T, "seed": 100 + i}) for i, (S, T) in enumerate(shapes)]

    def grads_of(net, ctc, b):
        net.SetSeqLengths(b.lens)
        out = net.Propagate(b.feats)
        d = ctc.EvalParallel(b.lens, out, b.labels)
        idf = CuMatrix(b.T * b.S, cfg["D"])
        net.BackpropagateNoUpdate(d, idf)
        return out.numpy(), idf.numpy(), net.GetGrads()

    fresh = []
    for b in batches:
        net = Net.from_layers(layers); ctc = Ctc()
        fresh.append(grads_of(net, ctc, b))
    net = Net.from_layers(layers); ctc = Ctc()
    order = [0, 1, 2, 3, 4, 1, 0, 3, 2]       # growing, then back to shorter ones on the grown buffers
    for i in order:
        got = grads_of(net, ctc, batches[i])
        for g, f in zip(got, fresh[i]):
            assert np.array_equal(g, f), (i, shapes[i])


def test_ctc_edge_cases(gpu):
    """Infeasible alignment (fewer frames than the labels need: ln p ~ -1e30 in the reference), a
    one-label utterance, and the expanded-label limit."""
    from eesen_amd.api import CuMatrix, Ctc, EesenError
    from oracle import net as onet
    rng = np.random.default_rng(1)
    S, T, K = 3, 6, 5
    lens = np.array([3, 6, 6], np.int32)
    labels = [np.array([2, 2, 2], np.int32),        # needs 5 frames (blanks between repeats), has 3: infeasible
              np.array([1], np.int32), np.array([3, 4, 1], np.int32)]
    x = rng.standard_normal((T * S, K)).astype(np.float32)
    p = np.exp(x - x.max(1, keepdims=True)); p = (p / p.sum(1, keepdims=True)).astype(np.float32)
    ids = np.concatenate(labels); off = np.concatenate([[0], np.cumsum([len(l) for l in labels])]).astype(np.int32)
    want = onet.ctc_eval_parallel(p, T, S, lens, ids, off, "f32")
    ctc = Ctc()
    diff = ctc.EvalParallel(lens, CuMatrix.from_numpy(p), labels).numpy()
    assert ctc.pzx[0] < -1e29 and want["pzx"][0] < -1e29
    assert rel_err(ctc.pzx[1:], want["pzx"][1:]) < 1e-6
    assert np.all(np.isfinite(diff)) and rel_err(diff, want["diff"]) < TOL
    with pytest.raises(EesenError, match="above 4096"):
        ctc.EvalParallel([4200], CuMatrix(4200, K), [np.ones(2100, np.int32)])


def _long_label_case(S, T, K, U, seed):
    """Character-style targets on long utterances: the LAST sequence has exactly U labels on T frames (the lattice the padded row is
    sized for), the others fewer labels on shorter utterances; ~15 % adjacent repeats."""
    rng = np.random.default_rng(seed)
    lens = np.sort(rng.integers(int(0.85 * T), T + 1, size=S)).astype(np.int32)
    lens[-1] = T
    logits = rng.standard_normal((T * S, K)).astype(np.float32) * 1.5
    probs = np.exp(logits - logits.max(1, keepdims=True)); probs /= probs.sum(1, keepdims=True)
    labels = []
    for s in range(S):
        u = U if s == S - 1 else int(rng.integers(U // 2, int(0.8 * U)))      # (feasible on the shorter utterances, repeats included)
        lab = rng.integers(1, K, size=u).astype(np.int32)
        for i in range(1, u):
            if rng.random() < 0.15: lab[i] = lab[i - 1]
        labels.append(lab)
    return lens, probs.astype(np.float32), labels


@pytest.mark.parametrize("S,T,K,U", [(3, 3000, 30, 600), (3, 3000, 30, 1000), (2, 3400, 12, 1600), (2, 2900, 8, 2047)])
def test_ctc_lattices_above_1024_positions(gpu, S, T, K, U):
    """The reference takes any label length (ctc-loss.cc:116-129; launch shape cuda-matrix.cc:868-898): character targets on a 35 s
    utterance exceed 511 labels.  Lattices of 1025 .. 4096 positions run as 2 / 4 wavefronts of one workgroup per lattice, the two
    values that cross a wave boundary handed over through LDS every step (csrc/ctc.hip): held against the oracle exactly like the
    one-wave lattices of test_ctc_vs_oracle -- the -1e30 sentinel pattern exact, the values to fp32 round-off (|alpha| reaches 1e4
    here: a relative bar), ln p, the gradient against the fp32 oracle and the fp64 arbiter with the oracle's own fp32 floor."""
    from eesen_amd.api import CuMatrix, Ctc
    from oracle import net as onet
    lens, probs, labels = _long_label_case(S, T, K, U, seed=U)
    ids = np.concatenate(labels); off = np.concatenate([[0], np.cumsum([len(l) for l in labels])]).astype(np.int32)
    want = onet.ctc_eval_parallel(probs, T, S, lens, ids, off, "f32")
    ctc = Ctc()
    dprob = CuMatrix.from_numpy(probs)
    diff = ctc.EvalParallel(lens, dprob, labels).numpy()
    ctc._rows = T * S
    a, b = ctc.alpha_beta()
    assert a.shape[1] == 2 * U + 1 > 1024
    for got, ref in ((a, want["alpha"]), (b, want["beta"])):
        assert np.array_equal(got == -1e30, ref == -1e30)
        m = ref != -1e30
        assert np.max(np.abs(got[m] - ref[m]) / np.maximum(1.0, np.abs(ref[m]))) < 5e-6
    assert np.all(np.isfinite(ctc.pzx)) and rel_err(ctc.pzx, want["pzx"]) < 2e-6
    arb = onet.ctc_eval_parallel(probs, T, S, lens, ids, off, "f64")
    floor = rel_err(want["diff"], arb["diff"])
    assert rel_err(diff, want["diff"]) < max(TOL, 3 * floor) and rel_err(diff, arb["diff"]) < max(TOL, 3 * floor)
    assert np.all(diff[~valid_mask(lens, T, S)] == 0)
    ne, nr = ctc.ErrorRateMSeq(lens, dprob, labels)
    assert (ne, nr) == onet.ctc_error_rate_mseq(probs, T, S, lens, ids, off)


@pytest.mark.parametrize("S,T,K,U,waves", [(4, 700, 20, 200, (1, 2, 4)), (3, 900, 25, 450, (1, 2, 4, 8)), (2, 1500, 16, 700, (2, 4, 8, 16)),
                                           (2, 2400, 10, 1100, (4, 8, 16))])
def test_ctc_multi_wave_sweep_is_bit_identical(gpu, S, T, K, U, waves, monkeypatch):
    """The arithmetic of a lattice position does not depend on which lane of which wavefront owns it: the sweep as n wavefronts per
    lattice (EESEN_CTC_WAVES=n, read when the Ctc is created) must reproduce the default kernel's alpha, beta, ln p and gradient BIT
    FOR BIT -- at 512 and 1024 positions the one-wave kernel (n = 1) among them, above that every instantiation that covers the row."""
    from eesen_amd.api import CuMatrix, Ctc
    lens, probs, labels = _long_label_case(S, T, K, U, seed=7 * U)
    dprob = CuMatrix.from_numpy(probs)

    def run():
        ctc = Ctc()
        diff = ctc.EvalParallel(lens, dprob, labels).numpy()
        ctc._rows = T * S
        a, b = ctc.alpha_beta()
        return a, b, ctc.pzx.copy(), diff

    base = run()
    assert np.all(np.isfinite(base[2])) and np.abs(base[3]).max() > 0
    for w in waves:
        monkeypatch.setenv("EESEN_CTC_WAVES", str(w))
        got = run()
        monkeypatch.delenv("EESEN_CTC_WAVES")
        for g, r in zip(got, base):
            assert np.array_equal(g, r), w


def test_recovery_from_a_timed_out_persistent_kernel(gpu, monkeypatch, capfd):
    """A cooperative recurrence kernel whose bounded spin gives up (here: a spin bound of one poll) must not corrupt the
    model: the update of that step is skipped ON THE DEVICE, the handle falls back to the per-step kernels with a warning,
    and training continues exactly as a per-step run from the same parameters would."""
    from eesen_amd.api import Net, Ctc
    cfg = synth.config("small_bi")
    layers = synth.make_model(max_grad=1.0, **cfg)
    batch = synth.make_batch(**cfg)

    def step(net, ctc):
        net.SetSeqLengths(batch.lens)
        out = net.Propagate(batch.feats)
        diff = ctc.EvalParallel(batch.lens, out, batch.labels)
        net.Backpropagate(diff)

    monkeypatch.setenv("EESEN_SPIN_LIMIT", "1")
    bad = Net.from_layers(layers); bad.SetTrainOptions(1e-3, 0.9); cb = Ctc()
    monkeypatch.delenv("EESEN_SPIN_LIMIT")
    p0 = bad.GetParams()
    step(bad, cb)                                  # the persistent kernels give up: garbage gradients, update skipped
    assert np.array_equal(bad.GetParams(), p0)     # (GetParams synchronises: the host notices and falls back here)
    bad.RecurrenceInfo()
    assert bad.recoveries == 1
    assert "NOT applied" in capfd.readouterr().err
    step(bad, cb)
    assert bad.RecurrenceInfo()["fwd_persistent"] == 0 and bad.recoveries == 1
    monkeypatch.setenv("EESEN_PERSISTENT", "0")
    ref = Net.from_layers(layers); ref.SetTrainOptions(1e-3, 0.9); cr = Ctc()
    step(ref, cr)
    assert np.array_equal(bad.GetParams(), ref.GetParams())


@pytest.mark.parametrize("K", [5000, 5121, 9000, 20480])
def test_ctc_with_thousands_of_classes(gpu, K):
    """Word / BPE-sized output layers: the gradient pass keeps 2 K floats per wave in LDS -- beyond the default 64 KB of dynamic LDS per
    workgroup at K ~ 2000 (the kernel then asks for up to 160 KB), four waves per workgroup up to 5120 classes, two up to 10240, one up to
    20480 (csrc/ctc.hip: ctc_error_diff); more than that is refused with a message."""
    from eesen_amd.api import CuMatrix, Ctc
    from eesen_amd._lib import EesenError
    from oracle import net as onet
    S, T = 3, 30
    lens, probs, labels = _random_ctc_case(S, T, K, 8, seed=99)
    if K == 20480:
        lens2, probs2, labels2 = _random_ctc_case(S, 4, K + 1, 2, seed=5)
        with pytest.raises(EesenError, match="more than 20480 classes"):
            Ctc().EvalParallel(lens2, CuMatrix.from_numpy(probs2), labels2)
    ids = np.concatenate(labels); off = np.concatenate([[0], np.cumsum([len(l) for l in labels])]).astype(np.int32)
    want = onet.ctc_eval_parallel(probs, T, S, lens, ids, off, "f32")
    ctc = Ctc()
    diff = ctc.EvalParallel(lens, CuMatrix.from_numpy(probs), labels).numpy()
    assert rel_err(ctc.pzx, want["pzx"]) < 1e-6 and rel_err(diff, want["diff"]) < TOL
