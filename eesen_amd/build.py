"""Builds eesen_amd/lib/libeesen_hip.so from eesen_amd/csrc with hipcc for gfx950 (in-tree, no JIT cache).

`python -m eesen_amd.build` or __graft_entry__.build().  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_DIR, "csrc")
LIBDIR = os.path.join(_DIR, "lib")
LIB = os.path.join(LIBDIR, "libeesen_hip.so")
SOURCES = ["gemm.hip", "lstm.hip", "lstm_persistent.hip", "ctc.hip", "optim.hip", "feeder.hip", "net.cpp", "ctc_host.cpp", "nnet_format.cpp", "capi.cpp", "comm.cpp"]
HEADERS = ["common.h", "guard.h", "kernels.h", "net.h", "handles.h", "tuning.h", os.path.join("..", "..", "include", "eesen_hip.h")]
FLAGS = (os.environ.get("EESEN_BUILD_DEFS", "").split()) + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


BINDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin")
TOOLS = {"train-ctc-parallel": "train_ctc_parallel.cc", "net-output-extract": "net_output_extract.cc"}


def csrc_digest() -> str:
    """sha1 over the library's sources (csrc/*, include/*.h): what the committed measurement records (profiles/pmc_traffic.json,
    profiles/parity_cfg*.json) are stamped with, so that a reader -- bench.py, on a GPU box that has no .git -- can tell whether
    they were taken on the tree that is running."""
    import hashlib
    h = hashlib.sha1()
    files = []
    for d in (CSRC, os.path.join(CSRC, "tools"), os.path.join(_DIR, "..", "include")):
        files += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith((".hip", ".cpp", ".h", ".cc", ".map"))]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    cc = hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s.replace(".", "_") + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [cc] + FLAGS + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=8) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn:
                print(warn, file=sys.stderr)
    if jobs or force or _stale(LIB, objs + [os.path.join(CSRC, "exports.map")]):
        run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs +
            ["-ldl", "-Wl,-Bsymbolic", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map")])
    # host-only C++ tools over the C-ABI (no HIP in these sources): the reference's trainer binary, natively
    os.makedirs(BINDIR, exist_ok=True)
    cxx = shutil.which("g++") or "g++"
    for name, src in TOOLS.items():
        exe = os.path.join(BINDIR, name)
        srcp = os.path.join(CSRC, "tools", src)
        if force or _stale(exe, [srcp, LIB, os.path.join(CSRC, "tools", "kaldi_tables.h"), os.path.join(CSRC, "tools", "feat_pipeline.h"), os.path.join(CSRC, "tools", "parse_options.h"),
                                 os.path.join(CSRC, "..", "..", "include", "eesen_hip.h"), os.path.join(CSRC, "..", "..", "include", "eesen_hip_info.h")]):
            run([cxx, "-O2", "-std=c++17", "-Wall", srcp, "-o", exe, "-L" + LIBDIR, "-leesen_hip", "-Wl,-rpath," + LIBDIR,
                 "-Wl,-rpath,$ORIGIN/../lib", "-Wl,--allow-shlib-undefined"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
