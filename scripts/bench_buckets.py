#!/usr/bin/env python3
"""BASELINE config 5 with bucketed variable-length padding: the recipes sort utterances by length (train_ctc_parallel.sh:84-89), so
consecutive minibatches are padded to different T_max.  Cycles minibatches of S utterances through the length buckets
(default 1000, 2000, 3000 frames) on the 6x1024 BiLSTM and reports padded frames/s over whole cycles, plus which recurrence
kernels ran.  Side measurement for profiles/; bench.py remains the contract."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eesen_amd import synth
from eesen_amd.api import Net, Ctc, CuMatrix


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg5")
    ap.add_argument("--buckets", default="1000,2000,3000")
    ap.add_argument("--cycles", type=int, default=2)
    a = ap.parse_args()
    cfg = synth.config(a.config)
    layers = synth.make_model(max_grad=50.0, **cfg)
    net = Net.from_layers(layers); net.SetTrainOptions(4e-5, 0.9); ctc = Ctc()
    batches = []
    for i, T in enumerate(int(x) for x in a.buckets.split(",")):
        b = synth.make_batch(**{**cfg, "T": T, "seed": 777 + i})
        batches.append((b, CuMatrix.from_numpy(b.feats), CuMatrix(b.T * b.S, cfg["K"])))

    def step(b, f, d):
        net.SetSeqLengths(b.lens)
        out = net.Propagate(f)
        ctc.EvalParallel(b.lens, out, b.labels, d, want_pzx=False)
        ctc.ErrorRateMSeq(b.lens, out, b.labels, deferred=True)
        net.Backpropagate(d)

    for b, f, d in batches:      # warm-up: allocations grow to the largest bucket
        step(b, f, d)
    net.Synchronize()
    per = {}
    t0 = time.perf_counter()
    for _ in range(a.cycles):
        for b, f, d in batches:
            t1 = time.perf_counter()
            step(b, f, d)
            net.Synchronize()
            per.setdefault(b.T, []).append(time.perf_counter() - t1)
    dt = time.perf_counter() - t0
    frames = a.cycles * sum(b.T * b.S for b, _, _ in batches)
    print(json.dumps({"config": a.config, "S": cfg["S"], "buckets": [b.T for b, _, _ in batches], "cycles": a.cycles,
                      "padded_frames_per_s": frames / dt, "ms_per_minibatch": {str(T): 1e3 * sum(v) / len(v) for T, v in per.items()},
                      "recurrence_kernels_last_step": net.RecurrenceInfo()}))


if __name__ == "__main__":
    main()
