#!/usr/bin/env python3
"""Microbenchmark of the GEMM on the shapes of the hot path (cfg2), in the three arithmetic modes (f32-input MFMA, 3-way bf16
split, two fp16 planes).  Prints fp32-equivalent TFLOP/s (2 M N K / time) per shape."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eesen_amd import _lib
from eesen_amd.api import CuMatrix

SHAPES = [  # name, a_kc, b_kc, M, N, K
    ("input->gates NT", 1, 1, 32000, 4096, 1024),
    ("input->gates L1 NT", 1, 1, 32000, 4096, 40),
    ("in_diff NN", 1, 0, 32000, 1024, 4096),
    ("Wx grad TN", 0, 0, 4096, 1024, 32000),
    ("Wm grad TN", 0, 0, 2048, 512, 32000),
    ("affine NT", 1, 1, 32000, 46, 1024),
    ("square NN 4096", 1, 0, 4096, 4096, 4096),
    ("one tile column NN (A read once)", 1, 0, 32000, 128, 4096),
]


def main():
    lib = _lib.load()
    rng = np.random.default_rng(0)
    only = os.environ.get("GEMM_BENCH_ONLY")
    for name, akc, bkc, M, N, K in SHAPES:
        if only and only not in name:
            continue
        ar, ac = (M, K) if akc else (K, M)
        br, bc = (N, K) if bkc else (K, N)
        A = CuMatrix.from_numpy(rng.uniform(-1, 1, (ar, ac)).astype(np.float32))
        B = CuMatrix.from_numpy(rng.uniform(-1, 1, (br, bc)).astype(np.float32))
        Cm = CuMatrix(M, N)
        for mode, mname in ((0, "f32-mfma"), (1, "bf16-split"), (2, "f16-planes")):
            lib.eesen_set_gemm_mode(mode)
            ms = C.c_float()
            _lib.check(lib.eesen_op_gemm_bench(0, akc, bkc, M, N, K, C.c_void_p(A.ptr), A.stride, C.c_void_p(B.ptr), B.stride,
                                               C.c_void_p(Cm.ptr), Cm.stride, 5, C.byref(ms)))
            print(f"{name:34s} {mname:10s} M={M:6d} N={N:5d} K={K:6d}  {ms.value:8.3f} ms  {2.0 * M * N * K / ms.value / 1e9:7.1f} TFLOP/s", flush=True)
        lib.eesen_set_gemm_mode(-1)


if __name__ == "__main__":
    main()
