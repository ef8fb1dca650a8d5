#!/usr/bin/env python3
"""Why is bench.py's in-process comparison leg (config.bf16_split_gemm) 4 ms slower than the same arithmetic as its own process?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eesen_amd import synth, _lib
from eesen_amd.api import Net, Ctc, CuMatrix
lib = _lib.load()
cfg = synth.config("cfg2"); layers = synth.make_model(max_grad=50.0, **cfg); batch = synth.make_batch(**cfg)
feats = CuMatrix.from_numpy(batch.feats); diff = CuMatrix(batch.T * batch.S, cfg["K"])
ctc = Ctc()
def run(tag, mode, prof, n=20):
    lib.eesen_set_gemm_mode(mode)
    net = Net.from_layers(layers); net.SetTrainOptions(4e-5, 0.9)
    if prof == "step": net.SetProfiling(True)
    elif prof == "acc": net.SetProfiling(True, accumulate=True)
    def step():
        net.SetSeqLengths(batch.lens); o = net.Propagate(feats)
        ctc.EvalParallel(batch.lens, o, batch.labels, diff, want_pzx=False); ctc.ErrorRateMSeq(batch.lens, o, batch.labels, deferred=True)
        net.Backpropagate(diff)
    for _ in range(3): step()
    net.Synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    net.Synchronize(); dt = (time.perf_counter() - t0) / n
    print(tag, "mode", mode, "profiling", prof, round(1e3 * dt, 2), "ms", flush=True)
    lib.eesen_set_gemm_mode(-1)
    return net
order = sys.argv[1] if len(sys.argv) > 1 else "a"
if order == "a":
    keep = run("first net", 2, "acc")
    run("second net", 1, "step"); run("third net", 1, "acc"); run("fourth net", 1, "off"); run("fifth net", 2, "step"); run("sixth", 2, "off")
else:
    run("first net", 1, "step"); run("second", 1, "off"); run("third", 2, "off")
