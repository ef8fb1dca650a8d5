"""Build (gcc) and load oracle/eesen_oracle.c in fp32 and fp64 — TEST INFRASTRUCTURE, not product code."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_DIR, "eesen_oracle.c")


def lib_path(prec: str) -> str:
    return os.path.join(_DIR, f"liboracle_{prec}.so")


def build(force: bool = False):
    for prec, ctype in (("f32", "float"), ("f64", "double")):
        out = lib_path(prec)
        if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(_SRC):
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=c99", "-fno-fast-math", "-ffp-contract=off",
                                   f"-DORC_REAL={ctype}", _SRC, "-o", out, "-lm"])


_LIBS = {}


def load(prec: str = "f32") -> C.CDLL:
    if prec not in _LIBS:
        build()
        lib = C.CDLL(lib_path(prec))
        assert lib.orc_sizeof_real() == (4 if prec == "f32" else 8)
        _LIBS[prec] = lib
    return _LIBS[prec]


def dtype(prec: str):
    return np.float32 if prec == "f32" else np.float64


def ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)
