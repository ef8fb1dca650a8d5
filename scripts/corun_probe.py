#!/usr/bin/env python3
"""What does a weight-gradient-sized GEMM cost a FORWARD recurrence that runs beside it, and what does the recurrence cost the GEMM?
(The side stream of the backward pass is known: GEMMs at a third of their rate, +1.0 ms per backward launch -- DESIGN.md section 5.)
Decides whether moving weight-gradient work from under the backward recurrences to under the NEXT step's forward recurrences can
pay (DESIGN.md section 10 "Deferred weight gradients").  cfg2 (default) or `--config cfg4 [--forward-bf16]`; the GEMM is the
W_x- / W_m-gradient shape (4096 x 1024 x 32000, TN: one direction's W_m gradient at H = 1024, both directions' W_x gradient at H = 512)
on its own non-blocking stream, launched back to back by a second host thread while the main thread times whole forward passes."""
import ctypes as C
import json
import sys
import threading
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eesen_amd import _lib, synth                    # noqa: E402
from eesen_amd.api import Net, Ctc, CuMatrix, check   # noqa: E402

lib = _lib.load()
hip = C.CDLL("libamdhip64.so")
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cfg2")
ap.add_argument("--forward-bf16", action="store_true")
ap.add_argument("--caps", default="48,16,0")
args = ap.parse_args()
cfg = synth.config(args.config)
layers = synth.make_model(max_grad=50.0, **cfg); batch = synth.make_batch(**cfg)
feats = CuMatrix.from_numpy(batch.feats)
diff = CuMatrix(batch.T * batch.S, cfg["K"])
rows = batch.T * batch.S
M, N, K = 4096, 1024, rows
A = CuMatrix(K, M, zero=False); B = CuMatrix(K, N, zero=False); Cm = CuMatrix(M, N, zero=False)
rng = np.random.default_rng(1)
for m in (A, B):   # random operands: the clock under a GEMM depends on the data (CDNA4 guide, DVFS)
    buf = rng.standard_normal((m.rows, m.stride)).astype(np.float32)
    check(lib.eesen_dev_copy(0, C.c_void_p(m.ptr), buf.ctypes.data_as(C.c_void_p), buf.nbytes, 1))
st = C.c_void_p()
assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0     # hipStreamNonBlocking: the Net's main stream
lo, hi = C.c_int(), C.c_int()
assert hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi)) == 0
sg = C.c_void_p()
assert hip.hipStreamCreateWithPriority(C.byref(sg), 1, lo.value) == 0   # the foreign GEMMs: lowest priority, like the library's side stream
net = Net.from_layers(layers, stream=st.value); net.SetTrainOptions(4e-5, 0.9)
net.SetForwardPrecision(1 if args.forward_bf16 else 0)
ctc = Ctc(stream=st.value)
WS = CuMatrix(1, 16 << 20, zero=False)    # split-K slabs


def launch(cap_kb):
    check(lib.eesen_op_gemm_async(0, sg, 0, 0, M, N, K, C.c_void_p(A.ptr), A.stride, C.c_void_p(B.ptr), B.stride, C.c_void_p(Cm.ptr), Cm.stride,
                                  C.c_void_p(WS.ptr), 16 << 20, cap_kb * 1024))


def sync_g():
    assert hip.hipStreamSynchronize(sg) == 0


def gemm_alone(cap_kb, n=20):
    launch(cap_kb); sync_g()
    t0 = time.perf_counter()
    for _ in range(n):
        launch(cap_kb)
    sync_g()
    return (time.perf_counter() - t0) / n * 1e3


def fwd():
    net.SetSeqLengths(batch.lens)
    out = net.Propagate(feats)
    net.Synchronize()
    return out


def bwd(out):
    ctc.EvalParallel(batch.lens, out, batch.labels, diff, want_pzx=False)
    net.BackpropagateNoUpdate(diff)
    net.Synchronize()


def timed(fn, n):
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3


def beside(fn, cap_kb, seconds=0.6):
    """fn() over and over while a second host thread keeps at most two of the foreign GEMMs queued on the low-priority stream."""
    stop = [False]; done = [0]

    def spin():
        while not stop[0]:
            launch(cap_kb); launch(cap_kb); sync_g(); done[0] += 2
    th = threading.Thread(target=spin); th.start()
    time.sleep(0.03)
    d0 = done[0]; t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        fn(); n += 1
    dt = time.perf_counter() - t0; d1 = done[0]
    stop[0] = True; th.join()
    return dt * 1e3 / n, dt * 1e3 / max(1, d1 - d0), (d1 - d0) / n


for _ in range(3):
    bwd(fwd())
res = {"config": args.config, "forward": "bf16" if args.forward_bf16 else "f32",
       "gemm": "weight-gradient shape %d x %d x %d (TN), %.0f GFLOP" % (M, N, K, 2.0 * M * N * K / 1e9)}
res["fwd_alone_ms"] = timed(fwd, 10)
out = fwd()
res["bwd_alone_ms"] = timed(lambda: bwd(out), 5)
for cap in [int(c) for c in args.caps.split(",")]:      # extra LDS > 0 selects the 128 x 128 flavour and caps its workgroups per CU: 48 KB = one (the side stream's setting), 16 KB = two; 0 = the 256 x 256 flavour, uncapped
    r = {"gemm_alone_ms": gemm_alone(cap)}
    for k, fn in (("fwd", fwd), ("bwd", lambda: bwd(out))):
        pass_ms, gemm_ms, per_pass = beside(fn, cap)
        r[k] = {"pass_ms": pass_ms, "gemm_ms": gemm_ms, "gemms_per_pass": per_pass,
                "growth_per_ms_of_standalone_gemm_hidden": (pass_ms - res[f"{k}_alone_ms"]) / max(1e-9, per_pass * r["gemm_alone_ms"])}
    res[f"cap_{cap}KB"] = r
res["note"] = ("forward pass = the recurrences + input GEMMs of the configuration; backward = CTC + the recurrences + all gradient GEMMs, its own side stream "
               "included; the foreign GEMMs run on a third, lowest-priority stream, at most two queued")
print(json.dumps(res))
