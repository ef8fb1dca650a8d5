#!/usr/bin/env python3
"""Probe of the two-plane GEMM on operands split AHEAD of the call (EESEN_GEMM_PRE=1: B as planes, 2: A and B):
(i) every shape of tests/test_gpu_gemm.py must come out bit-identical to the in-kernel split; (ii) TFLOP/s (fp32-equivalent) on the
shapes of the hot path, planes built outside the timed loop.  Usage: gemm_pre_probe.py check|bench [PRE]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BENCH = [  # name, a_kc, b_kc, M, N, K
    ("cfg2 input->gates NT", 1, 1, 32000, 4096, 1024),
    ("cfg2 in_diff NN", 1, 0, 32000, 1024, 4096),
    ("cfg2 Wx grad TN", 0, 0, 4096, 1024, 32000),
    ("cfg2 Wm grad TN", 0, 0, 2048, 512, 32000),
    ("cfg4 input->gates NT", 1, 1, 32000, 8192, 512),
    ("cfg4 in_diff NN", 1, 0, 32000, 512, 8192),
    ("cfg4 Wx grad TN", 0, 0, 8192, 512, 32000),
    ("cfg4 Wm grad TN", 0, 0, 4096, 1024, 32000),
    ("cfg5 input->gates NT", 1, 1, 192000, 8192, 2048),
    ("cfg5 in_diff NN", 1, 0, 192000, 2048, 8192),
    ("cfg5 Wx grad TN", 0, 0, 8192, 2048, 192000),
    ("cfg5 Wm grad TN", 0, 0, 4096, 1024, 192000),
]


def compute(pre):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_gemm import SHAPES
    from eesen_amd import _lib
    from eesen_amd.api import CuMatrix
    lib = _lib.load()
    lib.eesen_set_gemm_mode(2)
    out = {}
    for a_kc, b_kc, M, N, K in SHAPES + [(0, 0, 256, 256, 34), (1, 0, 300, 200, 50), (0, 0, 130, 70, 1026)]:
        rng = np.random.default_rng(M + 3 * N + 7 * K)
        A = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-7, 7, (M, K)))).astype(np.float32)
        B = (rng.standard_normal((K, N)) * np.exp(rng.uniform(-7, 7, (K, N)))).astype(np.float32)
        dA = CuMatrix.from_numpy(A if a_kc else np.ascontiguousarray(A.T))
        dB = CuMatrix.from_numpy(np.ascontiguousarray(B.T) if b_kc else B)
        dC = CuMatrix.from_numpy(np.zeros((M, N), np.float32))
        rc = lib.eesen_op_gemm(0, None, a_kc, b_kc, M, N, K, 1.0, C.c_void_p(dA.ptr), dA.stride, C.c_void_p(dB.ptr), dB.stride,
                               0.0, C.c_void_p(dC.ptr), dC.stride, None)
        assert rc == 0, lib.eesen_last_error()
        out[f"{a_kc}{b_kc}_{M}_{N}_{K}"] = dC.numpy()
    np.savez(f"/tmp/gemm_pre_{pre}.npz", **out)


def bench():
    from eesen_amd import _lib
    from eesen_amd.api import CuMatrix
    lib = _lib.load()
    lib.eesen_set_gemm_mode(2)
    rng = np.random.default_rng(0)
    only = os.environ.get("GEMM_BENCH_ONLY")
    for name, akc, bkc, M, N, K in BENCH:
        if only and only not in name:
            continue
        ar, ac = (M, K) if akc else (K, M)
        br, bc = (N, K) if bkc else (K, N)
        A = CuMatrix.from_numpy(rng.uniform(-1, 1, (ar, ac)).astype(np.float32))
        B = CuMatrix.from_numpy(rng.uniform(-1, 1, (br, bc)).astype(np.float32))
        Cm = CuMatrix(M, N)
        for pre in (0, 1, 2, 0):
            os.environ["EESEN_GEMM_PRE"] = str(pre)
            ms = C.c_float()
            _lib.check(lib.eesen_op_gemm_bench(0, akc, bkc, M, N, K, C.c_void_p(A.ptr), A.stride, C.c_void_p(B.ptr), B.stride,
                                               C.c_void_p(Cm.ptr), Cm.stride, int(os.environ.get('GEMM_ITERS', '30')), C.byref(ms)))
            print(f"{name:24s} pre={pre} M={M:6d} N={N:5d} K={K:6d}  {ms.value:8.3f} ms  {2.0 * M * N * K / ms.value / 1e9:7.1f} TF", flush=True)
        del A, B, Cm


if __name__ == "__main__":
    what = sys.argv[1]
    if what == "compute":
        compute(int(sys.argv[2]))
    elif what == "check":
        for pre in (0, 1, 2):
            env = dict(os.environ, EESEN_GEMM_PRE=str(pre))
            subprocess.run([sys.executable, os.path.abspath(__file__), "compute", str(pre)], env=env, check=True)
        ref = np.load("/tmp/gemm_pre_0.npz")
        for pre in (1, 2):
            got = np.load(f"/tmp/gemm_pre_{pre}.npz")
            for k in ref.files:
                same = np.array_equal(ref[k], got[k])
                print(f"pre={pre} {k}: {'bit-identical' if same else 'DIFFERS max ' + str(float(np.max(np.abs(ref[k] - got[k]))))}", flush=True)
    else:
        bench()
