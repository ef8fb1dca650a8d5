#!/usr/bin/env python3
"""Soak: N training steps of a configuration back to back; reports the step time and whether the persistent recurrence kernels
ever timed out and fell back (Net.recoveries must stay 0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eesen_amd import synth
from eesen_amd.api import Net, Ctc, CuMatrix

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cfg = synth.config(cfg_name)
if len(sys.argv) > 3:
    cfg["S"] = int(sys.argv[3])      # e.g. `soak.py cfg2 300 64`: --num-sequence 64 (two forward workgroups per CU, two-tile backward kernel)
layers = synth.make_model(max_grad=50.0, **cfg); batch = synth.make_batch(**cfg)
net = Net.from_layers(layers); net.SetTrainOptions(4e-5, 0.9)
ctc = Ctc()
feats = CuMatrix.from_numpy(batch.feats); diff = CuMatrix(batch.T * batch.S, cfg["K"])
def step():
    net.SetSeqLengths(batch.lens)
    out = net.Propagate(feats)
    ctc.EvalParallel(batch.lens, out, batch.labels, diff, want_pzx=False)
    net.Backpropagate(diff)
for _ in range(3): step()
net.Synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
net.Synchronize(); dt = time.perf_counter() - t0
info = net.RecurrenceInfo()
print(f"{cfg_name}: {steps} steps, {1e3 * dt / steps:.2f} ms/step, recurrence kernels {info}, recoveries {net.recoveries}")
assert net.recoveries == 0 and info["fwd_persistent"] == info["lstm_layers"] == info["bwd_persistent"]
