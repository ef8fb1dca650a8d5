#!/usr/bin/env python3
"""CTC lattice sweep: one wavefront per lattice against 2 / 4 / 8 wavefronts of one workgroup per lattice (EESEN_CTC_WAVES, csrc/ctc.hip),
at BASELINE config 5's CTC shape (S = 64 utterances, T = 3000, K = 51, 300 labels: L' = 601 in rows of 1024 positions) and at a
512-position shape.  Prints one JSON object: per setting the sweep's ms and us per lattice step, the bulk pass's ms."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eesen_amd.api import Ctc, CuMatrix   # noqa: E402

res = {}
for name, (S, T, K, U) in {"cfg2_ctc": (32, 1000, 46, 100), "cfg5_ctc": (64, 3000, 51, 300), "L401": (64, 2000, 46, 200), "L1201": (32, 3000, 46, 600), "L2401": (16, 3200, 30, 1200)}.items():
    rng = np.random.default_rng(5)
    lens = np.sort(rng.integers(int(0.8 * T), T + 1, size=S)).astype(np.int32); lens[-1] = T
    x = rng.standard_normal((T * S, K)).astype(np.float32)
    p = np.exp(x - x.max(1, keepdims=True)); p = (p / p.sum(1, keepdims=True)).astype(np.float32)
    labels = [rng.integers(1, K, size=U if s == S - 1 else max(1, int(lens[s]) * U // T)).astype(np.int32) for s in range(S)]
    probs = CuMatrix.from_numpy(p)
    diff = CuMatrix(T * S, K)
    res[name] = {"S": S, "T": T, "K": K, "Lprime": 2 * U + 1}
    base = None
    for w in (0, 1, 2, 4, 8, 16):
        if w:
            os.environ["EESEN_CTC_WAVES"] = str(w)
        else:
            os.environ.pop("EESEN_CTC_WAVES", None)
        ctc = Ctc()
        ctc.EvalParallel(lens, probs, labels, diff, want_pzx=False)
        ctc.SetProfiling(True)
        n = 5
        for _ in range(n):
            ctc.EvalParallel(lens, probs, labels, diff, want_pzx=False)
        ph = ctc.PhaseTimes()
        d = diff.numpy()
        if base is None:
            base = d
        res[name][f"waves_{w}"] = {"sweep_ms": 1e3 * ph["alpha_beta"] / n, "us_per_lattice_step": 1e6 * ph["alpha_beta"] / n / T,
                                  "bulk_ms": 1e3 * ph["error_diff"] / n, "log_ms": 1e3 * ph["log"] / n, "identical_to_default": bool(np.array_equal(d, base))}
        del ctc
os.environ.pop("EESEN_CTC_WAVES", None)
print(json.dumps(res))
