"""One trainer step of THE REFERENCE ITSELF (oracle/_ref) at a BASELINE configuration's full size, and the compact
fixture of it that travels in tests/golden/ for boxes without the reference library.  TEST INFRASTRUCTURE.

    python -m oracle.fullsize            # (re)generate tests/golden/full_*.npz where /root/reference exists

The step is the one of /root/reference/src/netbin/train-ctc-parallel.cc:195-207 with lr = 1, momentum = 0,
<MaxGrad> 0, which turns the parameter delta into the gradient.  The reference's CTC exists
only as CUDA kernels; oracle/_ref runs their bodies per emulated thread (oracle/ref_build/ref_cuda_emul.cc).
A full step of cfg2 (S=32, T=1000, 4x512 BiLSTM) takes ~12 s on 16 BLAS threads.

The fixture keeps what a size-independent comparison needs: ln p per sequence, per-tensor gradient statistics
(max |g|, sum, sum |g|) plus every STRIDE-th element, and every ROW_STRIDE-th row of net_out / diff / in_diff.
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np

from eesen_amd import nnet_io, synth

STRIDE = 1009        # gradient sample: every 1009th element of the Net::GetParams-ordered vector (prime: hits every tensor and row phase)
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
}
# cases whose reference step is too long to repeat inside the default GPU suite: the committed fixture (made by this script from
# the reference) is the arbiter unless EESEN_FULLSIZE_LIVE=1
FIXTURE_FIRST = {"full_cfg5_b1000"}


def case(name: str):
    cfg_name, over = CASES[name]
    cfg = synth.config(cfg_name)
    cfg.update(over)
    return cfg, synth.make_model(**cfg), synth.make_batch(**cfg)


def reference_step(layers, batch, blas_threads: int = 0, diff_override=None) -> dict:
    """Runs the reference. Returns net_out, pzx, diff, in_diff, grads (Net::GetParams order), alpha-free (too large).
    diff_override: backpropagate THIS matrix instead of the reference CTC's own gradient (e.g. an fp64 evaluation of the CTC on the
    reference's probabilities: how far the reference's fp32 CTC round-off moves the reference's own gradients)."""
    from oracle import refbind
    if blas_threads <= 0:
        blas_threads = min(16, os.cpu_count() or 1)
    refbind.set_blas_threads(blas_threads)
    path = tempfile.mktemp(suffix=".nnet")
    nnet_io.write_nnet(path, layers, binary=True)
    try:
        ref = refbind.RefNet(path)
    finally:
        os.unlink(path)
    before = ref.get_params()
    ref.set_train_options(1.0, 0.0)
    ref.set_seq_lengths(batch.lens)
    net_out = ref.propagate(batch.feats)
    ctc = refbind.cuda_ctc_eval_parallel(net_out, batch.T, batch.S, batch.lens, batch.label_ids, batch.label_off)
    ne, nr = ref.error_rate_mseq(net_out, batch.T, batch.S, batch.lens, batch.label_ids, batch.label_off)
    in_diff = ref.backpropagate(ctc["diff"] if diff_override is None else np.ascontiguousarray(diff_override, np.float32), True)
    grads = before.astype(np.float64) - ref.get_params().astype(np.float64)    # lr = 1, momentum = 0, no clipping
    return dict(net_out=net_out, pzx=ctc["pzx"], diff=ctc["diff"], in_diff=in_diff, grads=grads.astype(np.float32),
                errors=(ne, nr))


def tensor_stats(layers, flat) -> np.ndarray:
    """[n_tensors x 3]: max |g|, sum g, sum |g| per parameter tensor, in Net::GetParams order."""
    out, i = [], 0
    for L in layers:
        for p in L["params"]:
            g = np.asarray(flat[i:i + p.size], np.float64)
            out.append([np.max(np.abs(g)), g.sum(), np.abs(g).sum()])
            i += p.size
    assert i == len(flat)
    return np.array(out)


def compact(layers, r: dict) -> dict:
    return dict(pzx=r["pzx"], grad_stats=tensor_stats(layers, r["grads"]), grad_sample=r["grads"][::STRIDE].copy(),
                net_out_rows=r["net_out"][::ROW_STRIDE].copy(), diff_rows=r["diff"][::ROW_STRIDE].copy(),
                in_diff_rows=r["in_diff"][::ROW_STRIDE].copy(), errors=np.array(r["errors"], np.int64),
                diff_absmax=np.array(np.max(np.abs(r["diff"]))), in_diff_absmax=np.array(np.max(np.abs(r["in_diff"]))))


def reference_floors(layers, batch, r: dict) -> dict:
    """What the reference's OWN fp32 CTC round-off does to the reference's results, measured by evaluating the CTC in fp64 on the
    reference's probabilities (oracle/eesen_oracle.c, f64 build) and backpropagating THAT through the reference: the distance
    of its fp32 `diff` to the fp64 one, and per gradient tensor (and for in_diff) the shift.  These are the floors below which
    no fp32 implementation with a different summation order can be expected to agree with the reference end to end; the
    fixture carries them so that the bars of the test are the same with and without the library on the box."""
    from oracle import net as onet
    from tests.util import rel_err, split_params
    arb = onet.ctc_eval_parallel(r["net_out"], batch.T, batch.S, batch.lens, batch.label_ids, batch.label_off, "f64")
    r64 = reference_step(layers, batch, diff_override=arb["diff"])
    fl = [rel_err(a, b) for (_, _, a), (_, _, b) in zip(split_params(layers, r["grads"]), split_params(layers, r64["grads"]))]
    return dict(floor_diff=np.array(rel_err(r["diff"], arb["diff"])), floor_grads=np.array(fl),
                floor_in_diff=np.array(rel_err(r["in_diff"], r64["in_diff"])), diff64_rows=arb["diff"][::ROW_STRIDE].astype(np.float32))


def main():
    from oracle import refbind
    assert refbind.build_if_possible(), "oracle/_ref could not be built (needs /root/reference)"
    import time
    for name in (sys.argv[1:] or list(CASES)):
        cfg, layers, batch = case(name)
        t0 = time.time()
        r = reference_step(layers, batch)
        c = compact(layers, r)
        t1 = time.time()
        c.update(reference_floors(layers, batch, r))
        np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **c)
        print(f"{name}: reference step {t1 - t0:.1f} s (+ {time.time() - t1:.1f} s for the fp64-CTC floors), sum ln p = "
              f"{r['pzx'].astype(np.float64).sum():.4f}, errors {r['errors']}, floors: diff {float(c['floor_diff']):.2e}, "
              f"gradient tensors up to {float(c['floor_grads'].max()):.2e}", flush=True)


if __name__ == "__main__":
    main()
