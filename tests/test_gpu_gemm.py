"""-m gpu: the three arithmetic modes of the dense GEMM (include/eesen_hip.h `eesen_set_gemm_mode`) against an fp64 product.

Mode 0 (f32-input MFMA) is an exact fp32 fmaf chain.  Mode 1 splits every fp32 operand into three bf16 terms and runs
six bf16 MFMA products with fp32 accumulation; its error bound is 2^-23 |a*b| per product, the class of ONE fp32 rounding.
Mode 2 (round 6) holds every operand as two fp16 planes (round to nearest at both levels: 2^-22 relative in the worst case), every row /
column scaled by a power of two so that its largest magnitude sits in fp16's top binades, and runs three fp16 MFMA products: <= 3 * 2^-22
|a*b| per product (the "3xTF32" arithmetic).
The test measures all three against fp64, normalised by sum_k |a||b| (the quantity round-off scales with), over every operand
layout, ragged shapes, split-K shapes and the bias / alpha / beta epilogue, and requires the plane modes to be as accurate
as the fp32 chain."""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = [(1, 1, 256, 256, 64), (1, 1, 1000, 2048, 1024), (1, 0, 700, 300, 4096), (0, 0, 2048, 512, 32000), (0, 0, 184, 40, 8000),
          (0, 1, 257, 130, 50), (1, 1, 33, 46, 1024), (1, 0, 640, 1024, 46), (0, 0, 128, 128, 17), (1, 1, 129, 127, 19),
          # whole 256 x 256 tiles, >= 16 of them: the eight-wave flavour of the split kernel (all four layouts, split-K, short K)
          (1, 1, 1024, 1024, 512), (1, 0, 2048, 512, 4096), (0, 0, 1024, 1024, 8000), (0, 1, 1280, 1024, 48), (1, 1, 4096, 256, 16),
          # short K, many outputs, every layout: where a PER-TENSOR scale of the fp16 planes showed (3x the fp32 chain's error: an element 2^-20 of
          # its tensor's largest meeting a large partner); the scales are per row of op(A) / column of op(B)
          (0, 0, 4096, 256, 16), (1, 0, 2048, 512, 16), (0, 1, 1024, 1024, 32)]


def _run(gpu, mode, a_kc, b_kc, A, B, C0, bias, alpha, beta):
    from eesen_amd.api import CuMatrix
    assert gpu.eesen_set_gemm_mode(mode) == 0
    M, K = A.shape; N = B.shape[1]
    dA = CuMatrix.from_numpy(A if a_kc else np.ascontiguousarray(A.T))
    dB = CuMatrix.from_numpy(np.ascontiguousarray(B.T) if b_kc else B)
    dC = CuMatrix.from_numpy(C0)
    db = CuMatrix.from_numpy(bias[None, :])
    rc = gpu.eesen_op_gemm(0, None, a_kc, b_kc, M, N, K, alpha, C.c_void_p(dA.ptr), dA.stride, C.c_void_p(dB.ptr), dB.stride,
                           beta, C.c_void_p(dC.ptr), dC.stride, C.c_void_p(db.ptr))
    assert rc == 0, gpu.eesen_last_error()
    return dC.numpy()


@pytest.fixture(scope="module")
def report():
    rows = []
    yield rows.append
    try:
        out = os.environ.get("EESEN_PARITY_OUT", os.path.join(ROOT, "gpurun_out"))
        os.makedirs(out, exist_ok=True)
        json.dump(rows, open(os.path.join(out, "gemm_accuracy.json"), "w"), indent=1)
    except OSError:
        pass


@pytest.mark.parametrize("a_kc,b_kc,M,N,K", SHAPES)
def test_split_mode_is_as_accurate_as_the_fp32_chain(gpu, report, a_kc, b_kc, M, N, K):
    rng = np.random.default_rng(M + 3 * N + 7 * K)
    # wide dynamic range on purpose: magnitudes over ~6 decades, so that the low-order split terms matter
    A = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-7, 7, (M, K)))).astype(np.float32)
    B = (rng.standard_normal((K, N)) * np.exp(rng.uniform(-7, 7, (K, N)))).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    alpha, beta = 0.75, 0.5
    try:
        got = {m: _run(gpu, m, a_kc, b_kc, A, B, C0, bias, alpha, beta) for m in (0, 1, 2)}
    finally:
        gpu.eesen_set_gemm_mode(-1)
    A64, B64 = A.astype(np.float64), B.astype(np.float64)
    want = alpha * (A64 @ B64) + beta * C0 + bias[None, :]
    scale = alpha * (np.abs(A64) @ np.abs(B64)) + np.abs(beta * C0) + np.abs(bias)[None, :]
    err = {m: float(np.max(np.abs(got[m] - want) / scale)) for m in (0, 1, 2)}
    report(dict(a_kc=a_kc, b_kc=b_kc, M=M, N=N, K=K, err_f32_mfma=err[0], err_bf16_split=err[1], err_f16_planes=err[2]))
    assert err[0] < 4e-6                      # fp32 chain vs fp64, normalised by sum |a||b|: round-off class (grows slowly with K)
    assert err[1] < max(1.5 * err[0], 2.4e-7), f"split {err[1]:.3g} vs fp32 chain {err[0]:.3g}"
    assert err[2] < max(1.5 * err[0], 4e-7), f"fp16 planes {err[2]:.3g} vs fp32 chain {err[0]:.3g}"


def test_fp16_planes_keep_what_fp16_range_would_lose(gpu):
    """Mode 2's operands are fp16 planes of the operand times a power of two.  What has to hold for "fp32-class": (i) an element 2^-32
    of its tensor's largest lands in fp16's DENORMAL range (2^-18 after the scale): the MFMA must multiply it exactly, not flush it;
    (ii) operands whose magnitudes are far outside fp16's range altogether (1e-30, 1e30) come back scaled exactly; (iii) an all-zero
    operand gives zeros (its scale saturates), and a tensor whose largest element is huge does not overflow an fp16 plane."""
    K = 64
    A = np.zeros((128, K), np.float32); B = np.zeros((K, 128), np.float32)
    A[:, 0] = 1.0                      # the tensor's largest: sets the scale (1 -> 2^14)
    A[:, 1] = 2.0 ** -32               # 2^-18 after the scale: an fp16 denormal (64 units of 2^-24)
    A[:, 2] = 1.0 + 2.0 ** -20         # needs the lo plane (hi = 1, lo = 2^-20 -> 2^-6 after the scale)
    A[:, 3] = 3.0 * 2.0 ** -36         # 3 * 2^-22 after the scale: the denormal range's last bits
    B[1, 0] = 1.0; B[2, 1] = 1.0; B[3, 2] = 1.0; B[0, 3] = 0.5
    C0 = np.zeros((128, 128), np.float32); bias = np.zeros(128, np.float32)
    try:
        got = _run(gpu, 2, 1, 0, A, B, C0, bias, 1.0, 0.0)
        assert np.all(got[:, 0] == np.float32(2.0 ** -32)), got[0, :4]      # (i)
        assert np.all(got[:, 1] == np.float32(1.0 + 2.0 ** -20))
        assert np.all(got[:, 2] == np.float32(3.0 * 2.0 ** -36))
        assert np.all(got[:, 3] == 0.5)
        rng = np.random.default_rng(5)
        for sa, sb in ((1e-30, 1e30), (1e30, 1e-30), (1e-34, 1.0), (1.0, 1e37 / K)):   # (ii)
            A2 = (rng.standard_normal((128, K)) * sa).astype(np.float32); B2 = (rng.standard_normal((K, 128)) * sb).astype(np.float32)
            want = A2.astype(np.float64) @ B2.astype(np.float64)
            got = _run(gpu, 2, 1, 0, A2, B2, C0, bias, 1.0, 0.0)
            den = np.abs(A2).astype(np.float64) @ np.abs(B2).astype(np.float64)
            assert np.all(np.isfinite(got)) and float(np.max(np.abs(got - want) / den)) < 4e-7, (sa, sb)
        got = _run(gpu, 2, 1, 0, np.zeros_like(A), B, C0, bias, 1.0, 0.0)               # (iii)
        assert np.all(got == 0.0)
    finally:
        gpu.eesen_set_gemm_mode(-1)


@pytest.mark.parametrize("rows,cols,ld", [(768, 1024, 1024), (1024, 40, 40), (600, 46, 48), (1600, 4096, 4096), (7, 13, 16), (32000, 2048, 2048), (240, 2048, 2048),
                                          (3, 16384, 16384), (5000, 4, 4), (32000, 8192, 8192), (100, 5000, 5000), (64, 4100, 4104)])
def test_operand_bounds_pass(gpu, rows, cols, ld):
    """amax_rows_cols (one pass: row maxima by one wave per row, column maxima through LDS atomic max + a fold over the blocks) against
    numpy: exact (a maximum of magnitudes involves no rounding), with NaN-free inputs spanning many decades, rows-only / columns-only /
    both, padded rows (ld > cols: the pad must not be read)."""
    from eesen_amd.api import CuMatrix
    rng = np.random.default_rng(rows + cols)
    m = (rng.standard_normal((rows, ld)) * np.exp(rng.uniform(-20, 20, (rows, ld)))).astype(np.float32)
    m[:, cols:] = 1e30                      # the pad columns hold garbage that would win every maximum
    m[rows // 2, :cols] = 0.0               # an all-zero row
    m[:, cols // 2] = 0.0                   # an all-zero column
    d = CuMatrix.from_numpy(m)
    assert d.stride == ld
    want_r = np.abs(m[:, :cols]).max(1); want_c = np.abs(m[:, :cols]).max(0)
    for do_r, do_c in ((1, 1), (1, 0), (0, 1)):
        r = CuMatrix(1, rows); c = CuMatrix(1, cols)
        rc = gpu.eesen_op_amax_rows_cols(0, C.c_void_p(d.ptr), rows, cols, ld, C.c_void_p(r.ptr) if do_r else None, C.c_void_p(c.ptr) if do_c else None)
        assert rc == 0, gpu.eesen_last_error()
        if do_r: assert np.array_equal(r.numpy()[0], want_r)
        if do_c: assert np.array_equal(c.numpy()[0], want_c)


def test_training_step_in_split_mode_meets_the_same_parity_bar(gpu):
    """One full step (small_bi) with every GEMM in split mode against the oracle, at the 1e-4 bar of the fp32 path, and the
    distance between the two modes' gradients."""
    from eesen_amd import synth
    from eesen_amd.api import Net, Ctc
    from oracle import net as onet
    from tests.util import rel_err
    cfg = synth.config("cfg2"); cfg.update(T=40, S=16, layers=2)
    layers = synth.make_model(**cfg)
    batch = synth.make_batch(**cfg)
    ora = onet.OracleNet(layers, "f32"); ora.set_train_options(1.0, 0.0)
    o = onet.train_step(ora, batch, "f32")
    grads = {}
    try:
        for mode in (0, 1, 2):
            gpu.eesen_set_gemm_mode(mode)
            net = Net.from_layers(layers); net.SetTrainOptions(1.0, 0.0); ctc = Ctc()
            net.SetSeqLengths(batch.lens)
            out = net.Propagate(batch.feats)
            diff = ctc.EvalParallel(batch.lens, out, batch.labels)
            net.BackpropagateNoUpdate(diff)
            grads[mode] = net.GetGrads()
            assert rel_err(ctc.pzx, o["pzx"]) < 1e-4 and rel_err(diff.numpy(), o["diff"]) < 1e-4
            assert rel_err(grads[mode], ora.fresh_grads_flat()) < 1e-4
    finally:
        gpu.eesen_set_gemm_mode(-1)
    assert rel_err(grads[1], grads[0]) < 2e-5
    assert rel_err(grads[2], grads[0]) < 2e-5


def test_bf16_forward_variant(gpu):
    """BASELINE config 4's "bf16 forward / fp32 CTC accumulate" (eesen_net_set_forward_precision): forward GEMM operands rounded
    to bf16, everything else fp32.  No reference counterpart exists (BaseFloat = float), so the statement is a MEASURED distance
    to the fp32 path at cfg4's layer width (1024 cells + 512-d projections), written to gpurun_out/bf16_forward.json:
    ln p within 1e-2 relative, softmax outputs within 5e-2 of their maximum, gradient tensors within 0.15 (max-norm relative;
    a bf16 operand carries 2^-9 = 2e-3 relative rounding, which the recurrence amplifies over the layers)."""
    from eesen_amd import synth
    from eesen_amd.api import Net, Ctc
    from tests.util import rel_err, valid_mask, split_params
    cfg = synth.config("cfg4"); cfg.update(T=120, S=16, layers=3)
    layers = synth.make_model(**cfg)
    batch = synth.make_batch(**cfg)
    res = {}
    for bf16 in (False, True):
        net = Net.from_layers(layers); net.SetTrainOptions(1.0, 0.0); ctc = Ctc()
        net.SetForwardPrecision(bf16)
        net.SetSeqLengths(batch.lens)
        out = net.Propagate(batch.feats)
        diff = ctc.EvalParallel(batch.lens, out, batch.labels)
        net.BackpropagateNoUpdate(diff)
        res[bf16] = (out.numpy(), ctc.pzx.copy(), net.GetGrads())
    vm = valid_mask(batch.lens, batch.T, batch.S)
    rep = dict(config="cfg4 width: 3 x 1024-cell BiLSTM + 512-d projections, S=16, T=120",
               ln_p=rel_err(res[True][1], res[False][1]), net_out=rel_err(res[True][0][vm], res[False][0][vm]),
               grads={f"L{li}.{nm}": rel_err(a, b) for (li, nm, a), (_, _, b) in zip(split_params(layers, res[True][2]), split_params(layers, res[False][2]))})
    try:
        out_dir = os.environ.get("EESEN_PARITY_OUT", os.path.join(ROOT, "gpurun_out"))
        os.makedirs(out_dir, exist_ok=True)
        json.dump(rep, open(os.path.join(out_dir, "bf16_forward.json"), "w"), indent=1)
    except OSError:
        pass
    assert not np.array_equal(res[True][0], res[False][0])          # the option does something
    assert rep["ln_p"] < 1e-2 and rep["net_out"] < 5e-2
    assert max(rep["grads"].values()) < 0.15, rep["grads"]
