"""One trainer step of THE REFERENCE ITSELF (oracle/_ref) at a BASELINE configuration's full size, and the compact
fixture of it that travels in tests/golden/ for boxes without the reference library.  TEST INFRASTRUCTURE.

    python -m oracle.fullsize            # (re)generate tests/golden/full_*.npz where /root/reference exists

The step is the one of /root/reference/src/netbin/train-ctc-parallel.cc:195-207 with lr = 1, momentum = 0,
<MaxGrad> 0, which turns the parameter delta into the gradient.  The reference's CTC exists
only as CUDA kernels; oracle/_ref runs their bodies per emulated thread (oracle/ref_build/ref_cuda_emul.cc).
A full step of cfg2 (S=32, T=1000, 4x512 BiLSTM) takes ~12 s on 16 BLAS threads.

The fixture keeps what a size-independent comparison needs: ln p per sequence, per-tensor gradient statistics
(max |g|, sum, sum |g|, sum g^2), a SAMPLE of the gradient (every STRIDE-th element plus every tensor of at most SMALL elements
whole), and every ROW_STRIDE-th row of net_out / diff / in_diff -- each of the samples twice: from the reference step, and from
the reference's backward pass on an fp64 evaluation of ITS OWN CTC (the floor its fp32 CTC round-off imposes), so that every
error metric of tests/util.py::err_metrics can be formed for "HIP vs reference" and for "the reference vs its own fp64 CTC" on
the same elements, with and without the reference library on the box.
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np

from eesen_amd import nnet_io, synth

STRIDE = 1009        # gradient sample: every 1009th element of the Net::GetParams-ordered vector (prime: hits every tensor and row phase)
SMALL = 8192         # ... plus every tensor of at most this many elements whole (biases, peepholes: a stride sample would hold 1-4 of them)
ROW_STRIDE = 97      # row sample of the [T*S x .] matrices
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# name -> (synth config name, overrides).  full_cfg2 is exactly what bench.py times (seed 777).
CASES = {
    "full_cfg2": ("cfg2", {}),
    "full_cfg4_layer": ("cfg4", dict(layers=1, proj=0)),     # one 1024-cell BiLSTM layer at T=1000: the wide persistent tiles
    # BASELINE.json configs[3]: 5 x 1024 BiLSTM with 512-d <AffineTransform> projections between the layers
    # (asr_egs/wsj/utils/model_topo.py:99-128), K = 51, S = 32, T = 1000 -- exactly what bench.py's cfg4 leg times
    "full_cfg4": ("cfg4", {}),
    # BASELINE.json configs[4] at its 1000-frame length bucket: 6 x 1024 BiLSTM, S = 64 utterances per GPU (two sequence windows
    # of the wide backward tile, the time-multiplexed forward kernel), K = 51
    "full_cfg5_b1000": ("cfg5", dict(T=1000)),
    # BASELINE.json configs[4] at its 3000-FRAME bucket (U = 300 labels: L' = 601, lattice values |alpha| ~ 3e3).  The reference
    # keeps four [(T+2)S x 7H] state buffers per layer (bilstm-layer.h:1131-1136): 6 layers x S = 64 x T = 3000 are 132 GB of
    # them, twice the authoring container's memory, so the bucket is pinned by two cuts that each keep one axis whole:
    #   _l2:  TWO layers at the full S = 64 (the time-multiplexed kernels with two sequence tiles per workgroup, the gate-gradient
    #         buffer beyond 2 GB: T*S*8H*4 B = 6.3 GB, the PL = 10 lattice kernel at S = 64);
    #   _s16: the full SIX-layer stack at S = 16 (how six 1024-cell layers carry the 3000-frame lattice's round-off down).
    "full_cfg5_b3000_l2": ("cfg5", dict(layers=2)),
    "full_cfg5_b3000_s16": ("cfg5", dict(S=16)),
    # round 5: the kernels narrow layers take at --num-sequence 64 (two forward workgroups per CU, two 4-sequence tiles per backward
    # workgroup, no side stream) -- BASELINE configs[1]'s net on 64 utterances ...
    "full_cfg2_s64": ("cfg2", dict(S=64)),
    # ... and the recipes' own width (asr_egs/wsj/utils/model_topo.py: 320 cells per direction on 120-d features) at full length:
    # the 4 x 32 backward tile where K = 4H does not fill the waves' chunk pairs (lstm_bwd_persistent_q4_kernel<6, 4>)
    "full_recipe320": ("cfg2", dict(H=320, D=120)),
}
# cases whose reference step is too long to repeat inside the default GPU suite: the committed fixture (made by this script from
# the reference) is the arbiter unless EESEN_FULLSIZE_LIVE=1
FIXTURE_FIRST = {"full_cfg5_b1000", "full_cfg5_b3000_l2", "full_cfg5_b3000_s16", "full_cfg3", "full_cfg2_s64", "full_recipe320"}
# cases that only fit the host's memory with the layer-by-layer backward of ref_driver.cc (ref_net_backpropagate_lowmem: the
# reference's own per-layer Backpropagate + Update in Net::Backpropagate's order, each layer's state buffers released after use)
LOWMEM = {"full_cfg5_b3000_l2", "full_cfg5_b3000_s16", "full_cfg3"}

# BASELINE.json configs[2]: the GLOBAL minibatch of the 8-GPU run -- 256 utterances, T = 1000, 4 x 512 -- as ONE reference process
# with --num-sequence = 256 ("N ranks x S == reference with --num-sequence = N*S").  The HIP side runs it
# as 8 shards of 32 utterances (parallel.shard_batch: the interleaved deal the ranks use) and sums the gradients.
CFG3 = ("cfg2", dict(S=256))
CFG3_WORLD = 8
# the throughput settings of the recipes (asr_egs/wsj/run_ctc_phn.sh:84-85, utils/model_topo.py:90): what momentum and clipping
# do to the SUMMED gradient over two steps
CFG3_LR, CFG3_MOMENTUM, CFG3_MAX_GRAD, CFG3_STEPS = 4e-5, 0.9, 50.0, 2


def case(name: str):
    cfg_name, over = CFG3 if name == "full_cfg3" else CASES[name]
    cfg = synth.config(cfg_name)
    cfg.update(over)
    return cfg, synth.make_model(**cfg), synth.make_batch(**cfg)


def _ref_net(layers):
    from oracle import refbind
    path = tempfile.mktemp(suffix=".nnet")
    nnet_io.write_nnet(path, layers, binary=True)
    try:
        return refbind.RefNet(path)
    finally:
        os.unlink(path)


def reference_step(layers, batch, blas_threads: int = 0, diff_override=None, lowmem: bool = False) -> dict:
    """Runs the reference. Returns net_out, pzx, diff, in_diff, grads (Net::GetParams order), alpha-free (too large).
    diff_override: backpropagate THIS matrix instead of the reference CTC's own gradient (e.g. an fp64 evaluation of the CTC on the
    reference's probabilities: how far the reference's fp32 CTC round-off moves the reference's own gradients)."""
    from oracle import refbind
    if blas_threads <= 0:
        blas_threads = min(16, os.cpu_count() or 1)
    refbind.set_blas_threads(blas_threads)
    ref = _ref_net(layers)
    before = ref.get_params()
    ref.set_train_options(1.0, 0.0)
    ref.set_seq_lengths(batch.lens)
    net_out = ref.propagate(batch.feats)
    ctc = refbind.cuda_ctc_eval_parallel(net_out, batch.T, batch.S, batch.lens, batch.label_ids, batch.label_off)
    ne, nr = ref.error_rate_mseq(net_out, batch.T, batch.S, batch.lens, batch.label_ids, batch.label_off)
    in_diff = ref.backpropagate(ctc["diff"] if diff_override is None else np.ascontiguousarray(diff_override, np.float32), True,
                                lowmem=lowmem)
    grads = before.astype(np.float64) - ref.get_params().astype(np.float64)    # lr = 1, momentum = 0, no clipping
    return dict(net_out=net_out, pzx=ctc["pzx"], diff=ctc["diff"], in_diff=in_diff, grads=grads.astype(np.float32),
                errors=(ne, nr))


def reference_training_steps(layers, batch, lr: float, momentum: float, steps: int, blas_threads: int = 0, lowmem: bool = False):
    """`steps` trainer steps of the reference on the SAME minibatch with the given options (the layers carry <MaxGrad>):
    the parameter vectors after each step (Net::GetParams order), theta_0 first."""
    from oracle import refbind
    refbind.set_blas_threads(blas_threads if blas_threads > 0 else min(16, os.cpu_count() or 1))
    ref = _ref_net(layers)
    ref.set_train_options(lr, momentum)
    out = [ref.get_params()]
    for _ in range(steps):
        ref.set_seq_lengths(batch.lens)
        net_out = ref.propagate(batch.feats)
        ctc = refbind.cuda_ctc_eval_parallel(net_out, batch.T, batch.S, batch.lens, batch.label_ids, batch.label_off)
        ref.backpropagate(ctc["diff"], False, lowmem=lowmem)
        out.append(ref.get_params())
    return out


def tensor_bounds(layers):
    """[(begin, end)] of every parameter tensor in the Net::GetParams-ordered flat vector."""
    out, i = [], 0
    for L in layers:
        for p in L["params"]:
            out.append((i, i + p.size)); i += p.size
    return out


def tensor_stats(layers, flat) -> np.ndarray:
    """[n_tensors x 4]: max |g|, sum g, sum |g|, sum g^2 per parameter tensor, in Net::GetParams order."""
    out = []
    for a, b in tensor_bounds(layers):
        g = np.asarray(flat[a:b], np.float64)
        out.append([np.max(np.abs(g)), g.sum(), np.abs(g).sum(), np.square(g).sum()])
    assert b == len(flat)
    return np.array(out)


def sample_index(layers) -> np.ndarray:
    """Indices of the gradient sample: every STRIDE-th element, and every tensor of at most SMALL elements whole."""
    bounds = tensor_bounds(layers)
    m = np.zeros(bounds[-1][1], bool)
    m[::STRIDE] = True
    for a, b in bounds:
        if b - a <= SMALL:
            m[a:b] = True
    return np.flatnonzero(m)


def compact(layers, r: dict) -> dict:
    return dict(pzx=r["pzx"], grad_stats=tensor_stats(layers, r["grads"]), grad_sample=r["grads"][sample_index(layers)].copy(),
                net_out_rows=r["net_out"][::ROW_STRIDE].copy(), diff_rows=r["diff"][::ROW_STRIDE].copy(),
                in_diff_rows=r["in_diff"][::ROW_STRIDE].copy(), errors=np.array(r["errors"], np.int64),
                diff_absmax=np.array(np.max(np.abs(r["diff"]))), in_diff_absmax=np.array(np.max(np.abs(r["in_diff"]))))


def reference_floors(layers, batch, r: dict, lowmem: bool = False) -> dict:
    """What the reference's OWN fp32 CTC round-off does to the reference's results, measured by evaluating the CTC in fp64 on the
    reference's probabilities (oracle/eesen_oracle.c, f64 build) and backpropagating THAT through the reference: the distance
    of its fp32 `diff` to the fp64 one, and per gradient tensor (and for in_diff) the shift.  These are the floors below which
    no fp32 implementation with a different summation order can be expected to agree with the reference end to end; the
    fixture carries them -- as max-norm figures and as the SAMPLES of the fp64-CTC run, from which every other metric follows --
    so that the bars of the test are the same with and without the library on the box."""
    from oracle import net as onet
    from tests.util import rel_err, split_params
    arb = onet.ctc_eval_parallel(r["net_out"], batch.T, batch.S, batch.lens, batch.label_ids, batch.label_off, "f64")
    r64 = reference_step(layers, batch, diff_override=arb["diff"], lowmem=lowmem)
    fl = [rel_err(a, b) for (_, _, a), (_, _, b) in zip(split_params(layers, r["grads"]), split_params(layers, r64["grads"]))]
    return dict(floor_diff=np.array(rel_err(r["diff"], arb["diff"])), floor_grads=np.array(fl),
                floor_in_diff=np.array(rel_err(r["in_diff"], r64["in_diff"])), diff64_rows=arb["diff"][::ROW_STRIDE].astype(np.float32),
                in_diff64_rows=r64["in_diff"][::ROW_STRIDE].copy(), grad_sample64=r64["grads"][sample_index(layers)].copy(),
                grad_stats64=tensor_stats(layers, r64["grads"]))


def cfg3_training_layers(layers):
    """The same weights with the recipes' <MaxGrad> on every trainable layer."""
    out = []
    for L in layers:
        L = dict(L)
        if L["params"]:
            L["max_grad"] = CFG3_MAX_GRAD
        out.append(L)
    return out


def main():
    from oracle import refbind
    assert refbind.build_if_possible(), "oracle/_ref could not be built (needs /root/reference)"
    import time
    for name in (sys.argv[1:] or list(CASES) + ["full_cfg3"]):
        cfg, layers, batch = case(name)
        low = name in LOWMEM
        t0 = time.time()
        r = reference_step(layers, batch, lowmem=low)
        c = compact(layers, r)
        t1 = time.time()
        c.update(reference_floors(layers, batch, r, lowmem=low))
        t_floor = time.time() - t1
        extra = ""
        if name == "full_cfg3":     # two steps with momentum + <MaxGrad> on the summed gradient of the 256 utterances
            t2 = time.time()
            th = reference_training_steps(cfg3_training_layers(layers), batch, CFG3_LR, CFG3_MOMENTUM, CFG3_STEPS, lowmem=low)
            idx = sample_index(layers)
            for k in range(1, len(th)):
                d = th[k - 1].astype(np.float64) - th[k].astype(np.float64)
                c[f"delta{k}_sample"] = d[idx].astype(np.float32)
                c[f"delta{k}_stats"] = tensor_stats(layers, d)
            c["train_opts"] = np.array([CFG3_LR, CFG3_MOMENTUM, CFG3_MAX_GRAD, CFG3_STEPS])
            extra = f", {CFG3_STEPS} training steps {time.time() - t2:.1f} s"
        np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **c)
        print(f"{name}: reference step {t1 - t0:.1f} s (+ {t_floor:.1f} s for the fp64-CTC floors{extra}), sum ln p = "
              f"{r['pzx'].astype(np.float64).sum():.4f}, errors {r['errors']}, floors: diff {float(c['floor_diff']):.2e}, "
              f"gradient tensors up to {float(c['floor_grads'].max()):.2e}", flush=True)


if __name__ == "__main__":
    main()
