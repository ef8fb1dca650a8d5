"""-m gpu: the HIP path against THE REFERENCE ITSELF at the benchmarked size.

cfg2 exactly as bench.py runs it (S=32 utterances, T_max=1000, 4x512 BiLSTM, seed 777; lr=1 / momentum 0 / MaxGrad 0 so
that the parameter delta is the gradient): one trainer step of oracle/_ref (the reference's src/net + src/cpucompute
compiled unmodified, its CUDA CTC kernel bodies emulated per thread) against one step of libeesen_hip.so, both the
persistent recurrence and the per-step fallback, plus one 1024-cell layer at T=1000 (the wide persistent tiles).
When oracle/_ref is absent the compact fixture tests/golden/full_*.npz (made by `python -m oracle.fullsize` from the
reference) is the arbiter.

What can be held to 1e-4 and what cannot -- measured (profiles/parity_cfg2.json), not assumed:
  * forward (net_out on valid frames, ln p per sequence): 1e-4 relative; measured ~1e-6 / ~2e-7;
  * every parameter-gradient tensor, end to end (HIP forward + HIP CTC + HIP backward vs the reference): 1e-4 (measured 3-5e-5 at
    cfg2, up to 1.1e-4 on single peephole vectors of the 1024-cell layer), or 3x the distance the reference's OWN fp32 CTC
    round-off moves the reference's gradient of that tensor (reference backward on an fp64 CTC of its own probabilities);
  * the backward pass on its own (HIP backward fed with the REFERENCE's `diff`): in_diff and every gradient tensor 1e-4;
  * `diff` itself: gamma = exp(alpha + beta - ln p - ln y) carries the fp32 round-off of |alpha| ~ 1e3 in its exponent, so
    at T = 1000 ANY fp32 evaluation sits ~3.5e-3 (max-norm relative) from the fp64 value -- the reference's own CUDA
    arithmetic included (measured: reference 3.5e-3, HIP 3.6e-3, HIP vs reference 7e-4).  The bar for the CTC stage is
    therefore "as close to fp64 as the reference's fp32 is" (x1.5), and the end-to-end `in_diff`, which inherits that
    pointwise noise, must stay inside the reference's own floor.  The gradient tensors average it out over 32 000 frames.
"""
import json
import os
import time

import numpy as np
import pytest

from oracle import fullsize
from tests.util import rel_err, err_metrics, valid_mask, split_params

pytestmark = pytest.mark.gpu
TOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hip_step(layers, batch, persistent: bool, ref_diff=None, bf16_forward: bool = False):
    """One HIP step; with ref_diff a second backward pass runs on the reference's CTC gradient (stage isolation)."""
    from eesen_amd.api import Net, Ctc, CuMatrix
    old = os.environ.get("EESEN_PERSISTENT")
    os.environ["EESEN_PERSISTENT"] = "1" if persistent else "0"      # read when the Net is created
    try:
        net = Net.from_layers(layers)
    finally:
        if old is None:
            del os.environ["EESEN_PERSISTENT"]
        else:
            os.environ["EESEN_PERSISTENT"] = old
    net.SetTrainOptions(1.0, 0.0)
    net.SetForwardPrecision(bf16_forward)
    ctc = Ctc()
    net.SetSeqLengths(batch.lens)
    out = net.Propagate(batch.feats)
    diff = ctc.EvalParallel(batch.lens, out, batch.labels)
    errs = ctc.ErrorRateMSeq(batch.lens, out, batch.labels)
    info = net.RecurrenceInfo()
    bf16_layers = net.Bf16RecurrenceLayers()
    extra = {}
    if ref_diff is not None:     # backward only, on the reference's diff (before anything updates the weights)
        idf2 = CuMatrix(batch.T * batch.S, batch.feats.shape[1])
        net.BackpropagateNoUpdate(CuMatrix.from_numpy(ref_diff), idf2)
        extra = dict(bwd_in_diff=idf2.numpy(), bwd_grads=net.GetGrads())
    idf = CuMatrix(batch.T * batch.S, batch.feats.shape[1])
    net.BackpropagateNoUpdate(diff, idf)
    before = net.GetParams().astype(np.float64)
    grads = net.GetGrads()
    net.Update()
    net.Synchronize()
    delta = before - net.GetParams().astype(np.float64)     # what the reference exposes: theta_before - theta_after
    info.update(net.RecurrenceInfo())
    extra["bf16_layers"] = bf16_layers
    return dict(net_out=out.numpy(), pzx=ctc.pzx.copy(), diff=diff.numpy(), in_diff=idf.numpy(), grads=grads, delta=delta,
                errors=errs, recurrence=info, **extra)


_REF_CACHE = {}


def _reference(name):
    """Live reference step when oracle/_ref is on the box, else None (the fixture is then the arbiter).  Cases whose reference
    step takes minutes (fullsize.FIXTURE_FIRST) go by their committed fixture unless EESEN_FULLSIZE_LIVE=1."""
    if name not in _REF_CACHE:
        from oracle import refbind
        cfg, layers, batch = fullsize.case(name)
        r = None
        if refbind.available() and (name not in fullsize.FIXTURE_FIRST or os.environ.get("EESEN_FULLSIZE_LIVE") == "1"):
            t0 = time.time()
            r = fullsize.reference_step(layers, batch)
            r["seconds"] = time.time() - t0
        _REF_CACHE[name] = (cfg, layers, batch, r)
    return _REF_CACHE[name]


_REF64_CACHE = {}


def _reference_on_fp64_ctc(name, layers, batch, diff64):
    if name not in _REF64_CACHE:
        _REF64_CACHE[name] = fullsize.reference_step(layers, batch, diff_override=diff64)
    return _REF64_CACHE[name]


def _tensor_names(layers):
    out = []
    for li, L in enumerate(layers):
        if L["type"] in ("BiLstmParallel", "LstmParallel"):
            for d in (("fw", "bw") if L["type"] == "BiLstmParallel" else ("fw",)):
                out += [f"L{li}.{nm}_{d}" for nm in ("Wx", "Wm", "bias", "pi", "pf", "po")]
        elif L["type"] == "AffineTransform":
            out += [f"L{li}.W", f"L{li}.b"]
    return out


def _sample_slices(layers):
    """Per parameter tensor: (name, slice into the fixture's gradient sample), oracle/fullsize.py::sample_index order."""
    idx = fullsize.sample_index(layers)
    out = []
    for nm, (a, b) in zip(_tensor_names(layers), fullsize.tensor_bounds(layers)):
        lo, hi = np.searchsorted(idx, [a, b])
        out.append((nm, slice(int(lo), int(hi))))
    return out


def _fixture_grad_errors(layers, c, fx, key="grad"):
    """Per tensor, from the compact fixture: the sampled elements (every STRIDE-th of the flat gradient, small tensors whole) and
    the sums, each relative to the reference tensor's max |g| / sum |g|.  Same names as the live comparison."""
    out = {}
    st, fs = c[key + "_stats"], fx[key + "_stats"]
    for i, (nm, sl) in enumerate(_sample_slices(layers)):
        e = abs(st[i, 0] - fs[i, 0]) / fs[i, 0]
        e = max(e, abs(st[i, 1] - fs[i, 1]) / fs[i, 2], abs(st[i, 2] - fs[i, 2]) / fs[i, 2])
        if sl.stop > sl.start:
            e = max(e, float(np.max(np.abs(c[key + "_sample"][sl].astype(np.float64) - fx[key + "_sample"][sl]))) / fs[i, 0])
        out[nm] = float(e)
    return out


def _metric_table(layers, hip, ref=None, ref64=None, fx=None):
    """beside the max-norm ratio, the L2-relative error and the 99.9th-percentile elementwise relative error
    (tests/util.py::err_metrics) for `diff`, `in_diff` and every gradient tensor -- for HIP vs the reference AND for the reference's
    own fp32-vs-fp64-CTC floor, on the same elements: all of them with the live reference, the fixture's samples otherwise.
    Returns {quantity: {"hip_vs_reference": {...}, "reference_floor": {...}}}."""
    out = {}
    if ref is not None:
        pairs = [("diff", hip["diff"], ref["diff"], ref64["_diff64"]), ("in_diff", hip["in_diff"], ref["in_diff"], ref64["in_diff"])]
        for (li, nm, a), (_, _, b), (_, _, b64) in zip(split_params(layers, hip["grads"]), split_params(layers, ref["grads"]),
                                                        split_params(layers, ref64["grads"])):
            pairs.append((f"L{li}.{nm}", a, b, b64))
    else:
        c = fullsize.compact(layers, hip)
        pairs = [("diff", c["diff_rows"], fx["diff_rows"], fx["diff64_rows"]),
                 ("in_diff", c["in_diff_rows"], fx["in_diff_rows"], fx["in_diff64_rows"])]
        for nm, sl in _sample_slices(layers):
            pairs.append((nm, c["grad_sample"][sl], fx["grad_sample"][sl], fx["grad_sample64"][sl]))
    for nm, a, b, b64 in pairs:
        out[nm] = dict(hip_vs_reference=err_metrics(a, b), reference_floor=err_metrics(b, b64))
    return out


def _assert_metric_table(tab, factor=1.5, tail_factor=3.0, tol=TOL):
    """HIP may sit no further from the reference than `factor` x the reference sits from its own fp64-CTC evaluation in the
    L2-relative metric as well (the max-norm bars are the older, separate assertions).  The 99.9th percentile of the elementwise
    relative error is a TAIL statistic -- of a 2048-element bias vector it is the second-largest element's error, and two fp32
    evaluations of the same sum differ there by what one or two elements happen to round to (measured at cfg2: `L0.bias_fw` 4.0e-3
    against a floor of 2.3e-3, every other tensor 0.2-0.6 x its floor) -- so its bar is `tail_factor` x the floor."""
    bad = []
    for nm, t in tab.items():
        for m, f in (("l2", factor), ("p999", tail_factor)):
            if not t["hip_vs_reference"][m] <= max(tol, f * t["reference_floor"][m]):
                bad.append((nm, m, t["hip_vs_reference"][m], t["reference_floor"][m]))
    assert not bad, f"beyond the reference's own floor (x{factor} L2, x{tail_factor} 99.9th percentile): {bad}"


def _ctc_floor(net_out, batch, diff32):
    """fp64 arbiter of the CTC stage on the SAME probabilities: distance of an fp32 `diff` to it."""
    from oracle import net as onet
    arb = onet.ctc_eval_parallel(net_out, batch.T, batch.S, batch.lens, batch.label_ids, batch.label_off, "f64")
    return arb, rel_err(diff32, arb["diff"])


def _check(name, persistent, record, expect_all_persistent=True):
    cfg, layers, batch, ref = _reference(name)
    hip = _hip_step(layers, batch, persistent, ref["diff"] if ref is not None else None)
    if persistent and expect_all_persistent:   # the kernels the benchmark runs, not a silent per-step fallback
        ri = hip["recurrence"]
        assert ri["fwd_persistent"] == ri["lstm_layers"] == cfg["layers"] and ri["bwd_persistent"] == cfg["layers"], ri
    vm = valid_mask(batch.lens, batch.T, batch.S)
    rep = dict(case=name, persistent=persistent, S=batch.S, T=batch.T, reference="live oracle/_ref" if ref else "fixture tests/golden/%s.npz" % name)
    # the HIP gradient accessor and the black-box delta the reference exposes agree (lr = 1: delta = gradient)
    assert rel_err(hip["delta"], hip["grads"]) < 2e-6
    fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    if ref is not None:
        # the committed fixture is what this very reference code produced in the authoring container
        assert rel_err(fullsize.compact(layers, ref)["pzx"], fx["pzx"]) < 1e-5
        rep["reference_seconds"] = ref["seconds"]
        rep["ln_p"] = dict(hip=float(hip["pzx"].astype(np.float64).sum()), reference=float(ref["pzx"].astype(np.float64).sum()),
                           rel_err_per_sequence=rel_err(hip["pzx"], ref["pzx"]))
        rep["net_out_valid"] = rel_err(hip["net_out"][vm], ref["net_out"][vm])
        rep["in_diff"] = rel_err(hip["in_diff"], ref["in_diff"])
        rep["grads"] = {}
        for (li, nm, a), (_, _, b) in zip(split_params(layers, hip["grads"]), split_params(layers, ref["grads"])):
            rep["grads"][f"L{li}.{nm}"] = rel_err(a, b)
        # diff: the end-to-end distance and, per side, the distance to an fp64 CTC on that side's own probabilities
        arb_h, floor_h = _ctc_floor(hip["net_out"], batch, hip["diff"])
        arb_r, floor_r = _ctc_floor(ref["net_out"], batch, ref["diff"])
        rep["diff"] = dict(hip_vs_reference_fp32=rel_err(hip["diff"], ref["diff"]), hip_vs_fp64_on_hip_probs=floor_h,
                           reference_fp32_vs_fp64_on_reference_probs=floor_r,
                           fp64_hip_probs_vs_fp64_reference_probs=rel_err(arb_h["diff"], arb_r["diff"]),
                           frame_sums_hip_vs_reference=rel_err(hip["diff"].reshape(batch.T, batch.S, -1).sum(0),
                                                               ref["diff"].reshape(batch.T, batch.S, -1).sum(0)))
        rep["errors"] = dict(hip=list(hip["errors"]), reference=list(ref["errors"]))
        # how far the reference's own fp32 CTC round-off moves the reference's gradients: its backward pass on the fp64 CTC
        ref64 = _reference_on_fp64_ctc(name, layers, batch, arb_r["diff"])
        ref64["_diff64"] = arb_r["diff"]
        rep["metrics"] = _metric_table(layers, hip, ref, ref64)
        rep["reference_grads_fp32ctc_vs_fp64ctc"] = {}
        for (li, nm, a), (_, _, b) in zip(split_params(layers, ref["grads"]), split_params(layers, ref64["grads"])):
            rep["reference_grads_fp32ctc_vs_fp64ctc"][f"L{li}.{nm}"] = rel_err(a, b)
        # backward pass in isolation: HIP backward on the reference's own CTC gradient
        rep["backward_on_reference_diff"] = dict(in_diff=rel_err(hip["bwd_in_diff"], ref["in_diff"]), grads={})
        for (li, nm, a), (_, _, b) in zip(split_params(layers, hip["bwd_grads"]), split_params(layers, ref["grads"])):
            rep["backward_on_reference_diff"]["grads"][f"L{li}.{nm}"] = rel_err(a, b)
    else:
        c = fullsize.compact(layers, hip)
        rep["ln_p"] = dict(hip=float(hip["pzx"].astype(np.float64).sum()), reference=float(fx["pzx"].astype(np.float64).sum()),
                           rel_err_per_sequence=rel_err(hip["pzx"], fx["pzx"]))
        rs = fullsize.ROW_STRIDE
        rep["net_out_valid"] = rel_err(c["net_out_rows"][vm[::rs]], fx["net_out_rows"][vm[::rs]])
        rep["in_diff"] = float(np.max(np.abs(c["in_diff_rows"].astype(np.float64) - fx["in_diff_rows"])) / float(fx["in_diff_absmax"]))
        rep["grads"] = _fixture_grad_errors(layers, c, fx)
        rep["metrics"] = _metric_table(layers, hip, fx=fx)
        arb_h, floor_h = _ctc_floor(hip["net_out"], batch, hip["diff"])
        rep["diff"] = dict(hip_vs_reference_fp32=float(np.max(np.abs(c["diff_rows"].astype(np.float64) - fx["diff_rows"])) / float(fx["diff_absmax"])),
                           hip_vs_fp64_on_hip_probs=floor_h)
        if "floor_grads" in fx:   # the reference's own fp32-CTC floors, measured when the fixture was made (oracle/fullsize.py)
            rep["reference_grads_fp32ctc_vs_fp64ctc"] = dict(zip(rep["grads"].keys(), [float(x) for x in fx["floor_grads"]]))
            rep["diff"]["reference_fp32_vs_fp64_on_reference_probs"] = float(fx["floor_diff"])
            rep["reference_in_diff_fp32ctc_vs_fp64ctc"] = float(fx["floor_in_diff"])
        rep["errors"] = dict(hip=list(hip["errors"]), reference=[int(x) for x in fx["errors"]])
    record(rep)
    assert rep["ln_p"]["rel_err_per_sequence"] < TOL
    assert rep["net_out_valid"] < TOL
    floor_g = rep.get("reference_grads_fp32ctc_vs_fp64ctc", {})
    for k, v in rep["grads"].items():
        # within 1e-4, or within 3x the distance the reference's OWN fp32 CTC round-off moves that tensor; and never beyond 3e-4 unless
        # the reference's own floor for the tensor lies above that (the 5- and 6-layer 1024-cell stacks: floors up to 1.7e-3)
        fl = floor_g.get(k, 0.0)
        assert v < max(TOL, 3.0 * fl) and v < max(3 * TOL, fl), f"gradient tensor {k}: {v} (reference's own fp32-CTC floor {fl})"
    d = rep["diff"]
    if ref is not None:
        floor = d["reference_fp32_vs_fp64_on_reference_probs"]          # what the reference's own fp32 CTC arithmetic achieves
        assert d["hip_vs_fp64_on_hip_probs"] < max(TOL, 1.5 * floor)
        assert d["hip_vs_reference_fp32"] < max(TOL, floor) and rep["in_diff"] < max(TOL, floor)
        b = rep["backward_on_reference_diff"]
        assert b["in_diff"] < TOL
        for k, v in b["grads"].items():
            assert v < TOL, f"backward-only gradient tensor {k}: {v}"
        # greedy decode: identical up to argmax ties between probabilities that differ by ~1e-6 relative
        assert hip["errors"][1] == ref["errors"][1] and abs(hip["errors"][0] - ref["errors"][0]) <= 3
    elif "reference_fp32_vs_fp64_on_reference_probs" in d:   # fixture with the reference's floors: the same bars as live, minus the backward-only stage
        floor = d["reference_fp32_vs_fp64_on_reference_probs"]
        assert d["hip_vs_fp64_on_hip_probs"] < max(TOL, 1.5 * floor)
        assert d["hip_vs_reference_fp32"] < max(TOL, floor) and rep["in_diff"] < max(TOL, floor)
        assert rep["errors"]["hip"][1] == rep["errors"]["reference"][1] and abs(rep["errors"]["hip"][0] - rep["errors"]["reference"][0]) <= 3
    else:
        assert d["hip_vs_fp64_on_hip_probs"] < 6e-3 and d["hip_vs_reference_fp32"] < 6e-3 and rep["in_diff"] < 6e-3
        assert abs(rep["errors"]["hip"][0] - rep["errors"]["reference"][0]) <= 3
    if "metrics" in rep:
        _assert_metric_table(rep["metrics"])
    assert np.all(hip["diff"][~vm] == 0) and np.all(hip["in_diff"][~vm] == 0)


@pytest.fixture(scope="module")
def record():
    reps = []
    yield reps.append
    out_dir = os.environ.get("EESEN_PARITY_OUT", os.path.join(ROOT, "gpurun_out"))
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "parity_fullsize.json"), "w") as f:
            json.dump(reps, f, indent=1)
    except OSError:
        pass


def test_cfg2_full_length_against_reference(gpu, record):
    _check("full_cfg2", True, record)


def test_cfg2_full_length_per_step_kernels_against_reference(gpu, record):
    _check("full_cfg2", False, record)


def test_cfg2_net_at_num_sequence_64_against_reference(gpu, record):
    """Round 5: what narrow layers run at --num-sequence 64 -- the bf16-pipe forward tile as two workgroups per CU (census-backed),
    lstm_bwd_persistent_q4_kernel<8, 8>, gradient GEMMs off the side stream -- against the reference's own step on the same 64
    utterances of 1000 frames (fixture made by oracle/fullsize.py from the reference; EESEN_FULLSIZE_LIVE=1 repeats it live)."""
    _check("full_cfg2_s64", True, record)


def test_recipe_width_320_cells_full_length_against_reference(gpu, record):
    """Round 5: the recipes' own width, 4 x 320 cells on 120-d features, S = 32, T = 1000 -- the 4 x 32 backward tile with K = 4H not
    filling the waves' chunk pairs (lstm_bwd_persistent_q4_kernel<6, 4>), the narrow bf16-pipe forward tile at H = 320."""
    _check("full_recipe320", True, record)


def test_cfg4_wide_layer_full_length_against_reference(gpu, record):
    _check("full_cfg4_layer", True, record)


def test_cfg4_projected_stack_full_length_against_reference(gpu, record):
    """BASELINE.json configs[3] as bench.py's cfg4 leg runs it: 5 x 1024 BiLSTM with 512-d <AffineTransform> projections between
    the layers (asr_egs/wsj/utils/model_topo.py:99-128), K = 51, S = 32, T = 1000; the wide persistent tiles on every layer."""
    _check("full_cfg4", True, record)


def test_cfg5_1000_frame_bucket_against_reference(gpu, record):
    """BASELINE.json configs[4] at its 1000-frame bucket: 6 x 1024 BiLSTM, S = 64 utterances (the time-multiplexed forward kernel,
    two sequence windows of the wide backward tile).  The reference step takes minutes on the host, so the committed fixture --
    made from the reference by `python -m oracle.fullsize full_cfg5_b1000`, with the reference's own fp32-CTC floors -- is the
    arbiter unless EESEN_FULLSIZE_LIVE=1."""
    _check("full_cfg5_b1000", True, record)


def test_cfg4_bf16_forward_variant_distance_to_the_reference(gpu, record):
    """BASELINE.json configs[3]'s "bf16 forward / fp32 CTC accumulate" variant (eesen_net_set_forward_precision): its distance
    to THE REFERENCE at full size -- not to this library's own fp32 path -- goes on record (profiles/parity_cfg4.json) for both
    modes: 1 = forward GEMMs and the forward time recurrence on bf16 operands (lstm_fwd_persistent_bf16_kernel on all five
    layers, asserted), 2 = the GEMM operands only (round 3).  Bars: ln p within 2e-3 per sequence, every
    gradient tensor within 0.1 (max-norm relative), and the recurrence's rounding may cost at most 2x what the GEMM-only
    rounding already does."""
    name = "full_cfg4"
    cfg, layers, batch, ref = _reference(name)
    vm = valid_mask(batch.lens, batch.T, batch.S)
    fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    reps = {}
    for mode in (1, 2):
        hip = _hip_step(layers, batch, True, None, bf16_forward=mode)
        rep = dict(case=name + (" (bf16 forward: GEMMs + recurrence)" if mode == 1 else " (bf16 forward: GEMM operands only)"),
                   persistent=True, S=batch.S, T=batch.T, bf16_recurrence_layers=hip["bf16_layers"])
        assert hip["bf16_layers"] == (cfg["layers"] if mode == 1 else 0), hip["bf16_layers"]
        if ref is not None:
            rep["ln_p_rel_err_per_sequence"] = rel_err(hip["pzx"], ref["pzx"])
            rep["net_out_valid"] = rel_err(hip["net_out"][vm], ref["net_out"][vm])
            rep["grads"] = {f"L{li}.{nm}": rel_err(a, b) for (li, nm, a), (_, _, b) in zip(split_params(layers, hip["grads"]), split_params(layers, ref["grads"]))}
        else:
            c = fullsize.compact(layers, hip)
            rs = fullsize.ROW_STRIDE
            rep["ln_p_rel_err_per_sequence"] = rel_err(hip["pzx"], fx["pzx"])
            rep["net_out_valid"] = rel_err(c["net_out_rows"][vm[::rs]], fx["net_out_rows"][vm[::rs]])
            rep["grads"] = _fixture_grad_errors(layers, c, fx)
        rep["grads_worst"] = max(rep["grads"].values())
        record(rep)
        reps[mode] = rep
    for mode in (1, 2):
        assert reps[mode]["ln_p_rel_err_per_sequence"] < 2e-3 and reps[mode]["net_out_valid"] < 5e-2, reps[mode]
        assert reps[mode]["grads_worst"] < 0.1, reps[mode]["grads"]
    assert reps[1]["grads_worst"] < 2.0 * reps[2]["grads_worst"] and reps[1]["ln_p_rel_err_per_sequence"] < 2.0 * max(reps[2]["ln_p_rel_err_per_sequence"], 2.5e-4)


def test_cfg5_3000_frame_bucket_two_layers_at_full_batch_against_reference(gpu, record):
    """BASELINE.json configs[4] at its 3000-frame bucket, first cut (oracle/fullsize.py): two 1024-cell layers at the full S = 64 --
    U = 300 labels (L' = 601: the PL = 10 lattice kernel, |alpha| ~ 3e3), the time-multiplexed recurrence kernels with two
    sequence tiles per workgroup, the gate-gradient buffer beyond 2 GB (6.3 GB) -- against the reference's own step
    (bilstm-parallel-layer.h:97-206,422-602, ctc-loss.cc:101-194), not against this library's per-step twin."""
    _check("full_cfg5_b3000_l2", True, record)


def test_cfg5_3000_frame_bucket_six_layers_against_reference(gpu, record):
    """... second cut: the full six-layer stack at T = 3000 with S = 16 (the reference's state buffers for S = 64 are 132 GB)."""
    _check("full_cfg5_b3000_s16", True, record)


def _dev_read(ptr, n):
    import ctypes as C
    from eesen_amd import _lib
    out = np.empty(n, np.float32)
    _lib.check(_lib.load().eesen_dev_copy(0, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes, 2))
    return out


def _dev_write(ptr, a):
    import ctypes as C
    from eesen_amd import _lib
    a = np.ascontiguousarray(a, np.float32)
    _lib.check(_lib.load().eesen_dev_copy(0, C.c_void_p(ptr), a.ctypes.data_as(C.c_void_p), a.nbytes, 1))


def test_cfg3_eight_shards_equal_the_reference_on_the_global_minibatch(gpu, record):
    """BASELINE.json configs[2] / SURVEY.md section 8e's parity statement at full size: "N ranks x S == the reference with
    --num-sequence = N*S".  The arbiter is ONE reference process on the global minibatch of the 8-GPU run (256 utterances,
    T = 1000, 4 x 512; tests/golden/full_cfg3.npz, made by oracle/fullsize.py); the HIP side runs the eight shards of 32
    utterances the ranks would hold (parallel.shard_batch: the interleaved deal, each shard padded to its own T_max) one after
    the other on this GPU with the persistent kernels, and SUMS their fresh gradients in rank order -- what the all-reduce of
    csrc/comm.cpp does between Backpropagate and Update (replaces /root/reference/src/net/communicator.h:39-170).
      (i)  lr = 1, no momentum, no clipping: ln p of all 256 utterances, softmax outputs, `diff`, `in_diff`, and the summed
           gradient of every tensor against the reference's, with the reference's own fp32-CTC floors as in the other cases;
      (ii) the recipes' settings (lr 4e-5, momentum 0.9, <MaxGrad> 50: asr_egs/wsj/run_ctc_phn.sh:84-85), two steps: the summed
           gradient goes back into the Net's gradient buffer and eesen_net_update applies momentum and clipping to the SUM;
           the parameter deltas of both steps against the reference's."""
    from eesen_amd import parallel
    from eesen_amd.api import Net, Ctc, CuMatrix
    name = "full_cfg3"
    cfg, layers, batch = fullsize.case(name)
    fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    W = fullsize.CFG3_WORLD
    D, K = batch.feats.shape[1], cfg["K"]
    deal = parallel.deal_shards(batch.S, W)
    shards = [parallel.shard_batch(batch, r, W) for r in range(W)]
    assert all(sb.S == 32 for sb in shards)

    def shard_pass(net, ctc, sb, want_in_diff):
        net.SetSeqLengths(sb.lens)
        out = net.Propagate(sb.feats)
        diff = ctc.EvalParallel(sb.lens, out, sb.labels)
        idf = CuMatrix(sb.T * sb.S, D) if want_in_diff else None
        net.BackpropagateNoUpdate(diff, idf)
        return out, diff, idf

    # (i) gradients of the global minibatch
    net = Net.from_layers(layers)
    net.SetTrainOptions(1.0, 0.0)
    ctc = Ctc()
    g_sum = None
    pzx = np.zeros(batch.S, np.float32)
    glob = dict(net_out=np.zeros((batch.T, batch.S, K), np.float32), diff=np.zeros((batch.T, batch.S, K), np.float32),
                in_diff=np.zeros((batch.T, batch.S, D), np.float32))
    for r, sb in enumerate(shards):
        out, diff, idf = shard_pass(net, ctc, sb, True)
        g = net.GetGrads()
        g_sum = g if g_sum is None else g_sum + g          # fp32, rank order
        pzx[deal[r]] = ctc.pzx
        for key, m in (("net_out", out), ("diff", diff), ("in_diff", idf)):
            glob[key][: sb.T, deal[r], :] = m.numpy().reshape(sb.T, sb.S, -1)
    ri = net.RecurrenceInfo()
    assert ri["fwd_persistent"] == ri["lstm_layers"] == cfg["layers"] and ri["bwd_persistent"] == cfg["layers"], ri
    hip = dict(pzx=pzx, grads=g_sum, errors=(0, 0), **{k: v.reshape(batch.T * batch.S, -1) for k, v in glob.items()})
    vm = valid_mask(batch.lens, batch.T, batch.S)
    c = fullsize.compact(layers, hip)
    rs = fullsize.ROW_STRIDE
    rep = dict(case=name, persistent=True, S=batch.S, T=batch.T, shards=W, reference="fixture tests/golden/%s.npz (one reference process, --num-sequence 256)" % name)
    rep["ln_p"] = dict(hip=float(pzx.astype(np.float64).sum()), reference=float(fx["pzx"].astype(np.float64).sum()),
                       rel_err_per_sequence=rel_err(pzx, fx["pzx"]))
    rep["net_out_valid"] = rel_err(c["net_out_rows"][vm[::rs]], fx["net_out_rows"][vm[::rs]])
    rep["in_diff"] = float(np.max(np.abs(c["in_diff_rows"].astype(np.float64) - fx["in_diff_rows"])) / float(fx["in_diff_absmax"]))
    rep["diff"] = dict(hip_vs_reference_fp32=float(np.max(np.abs(c["diff_rows"].astype(np.float64) - fx["diff_rows"])) / float(fx["diff_absmax"])),
                       reference_fp32_vs_fp64_on_reference_probs=float(fx["floor_diff"]))
    rep["grads"] = _fixture_grad_errors(layers, c, fx)
    rep["reference_grads_fp32ctc_vs_fp64ctc"] = dict(zip(rep["grads"].keys(), [float(x) for x in fx["floor_grads"]]))
    rep["reference_in_diff_fp32ctc_vs_fp64ctc"] = float(fx["floor_in_diff"])
    rep["metrics"] = _metric_table(layers, hip, fx=fx)

    # (ii) two steps with momentum and clipping acting on the sum
    lr, mom, max_grad, steps = [float(x) for x in fx["train_opts"]]
    net2 = Net.from_layers(fullsize.cfg3_training_layers(layers))
    net2.SetTrainOptions(lr, mom)
    ctc2 = Ctc()
    gptr, gn = net2.grad_buffer()
    theta = [net2.GetParams()]
    rep["training"] = dict(lr=lr, momentum=mom, max_grad=max_grad, deltas=[])
    for k in range(1, int(steps) + 1):
        acc = None
        for sb in shards:
            shard_pass(net2, ctc2, sb, False)
            net2.Synchronize()
            g = _dev_read(gptr, gn)                   # the all-reduce payload itself (library layout), summed in rank order
            acc = g if acc is None else acc + g
        _dev_write(gptr, acc)
        net2.Update()
        net2.Synchronize()
        theta.append(net2.GetParams())
        d = theta[k - 1].astype(np.float64) - theta[k].astype(np.float64)
        # A parameter delta is lr x the clipped, momentum-folded gradient: the clip saturates its SIZE at lr x max_grad while the
        # round-off it carries is that of gradients up to ~1e4 (a quarter of the elements sit at the clip), so the error is put
        # against what it is an error OF -- lr x the largest summed-gradient element of the tensor -- and held to the gradient's bar
        # (twice that for step 2, whose buffer is 0.9 x step 1's plus a fresh gradient).
        d_h = d[fullsize.sample_index(layers)]
        d_r = fx[f"delta{k}_sample"].astype(np.float64)
        gmax = fx["grad_stats"][:, 0]
        e = {nm: float(np.max(np.abs(d_h[sl] - d_r[sl])) / (lr * gmax[i])) for i, (nm, sl) in enumerate(_sample_slices(layers)) if sl.stop > sl.start}
        at_clip = lambda a: float(np.mean(np.abs(a) >= lr * max_grad * (1 - 1e-4)))
        rep["training"]["deltas"].append(dict(step=k, per_tensor_error_over_lr_times_max_gradient=e, worst=max(e.values()),
                                              fraction_of_sampled_elements_at_the_clip=dict(hip=at_clip(d_h), reference=at_clip(d_r)),
                                              largest_delta=dict(hip=float(np.max(np.abs(d_h))), reference=float(np.max(np.abs(d_r))))))
    record(rep)

    assert rep["ln_p"]["rel_err_per_sequence"] < TOL and rep["net_out_valid"] < TOL
    floor_g = rep["reference_grads_fp32ctc_vs_fp64ctc"]
    for kk, v in rep["grads"].items():
        fl = floor_g[kk]
        assert v < max(TOL, 3.0 * fl) and v < max(3 * TOL, fl), f"summed gradient tensor {kk}: {v} (reference's own fp32-CTC floor {fl})"
    floor = rep["diff"]["reference_fp32_vs_fp64_on_reference_probs"]
    assert rep["diff"]["hip_vs_reference_fp32"] < max(TOL, floor) and rep["in_diff"] < max(TOL, floor)
    _assert_metric_table(rep["metrics"])
    for dd in rep["training"]["deltas"]:
        for kk, v in dd["per_tensor_error_over_lr_times_max_gradient"].items():
            assert v < dd["step"] * max(TOL, 3.0 * floor_g[kk]), f"step {dd['step']} parameter delta {kk}: {v} (gradient floor {floor_g[kk]})"
        c = dd["fraction_of_sampled_elements_at_the_clip"]
        assert abs(c["hip"] - c["reference"]) < 2e-3 and c["reference"] > 0.05          # momentum and <MaxGrad> really acted, alike
        assert abs(dd["largest_delta"]["hip"] - dd["largest_delta"]["reference"]) < 1e-4 * dd["largest_delta"]["reference"]
    assert np.all(hip["diff"][~vm] == 0) and np.all(hip["in_diff"][~vm] == 0)
