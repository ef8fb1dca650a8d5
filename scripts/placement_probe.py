#!/usr/bin/env python3
"""A Net's step time depends on the allocation history of its process.  Which phases move, and what kind of
history does it: a freed large Net in front (cfg4), allocation churn (many buffers of mixed sizes allocated and freed), a second Net alive
beside it?  The probe shape is the recipes' (4 x 320, D = 120, S = 10, T = 1000): a latency-bound chain, where the effect was largest.
Usage: placement_probe.py a|b|c|d|e"""
import gc, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eesen_amd import synth, _lib
from eesen_amd.api import Net, Ctc, CuMatrix
lib = _lib.load()
ctc = Ctc()

def timed(tag, base, over, n=10, keep=False):
    cfg = synth.config(base); cfg.update(over)
    layers = synth.make_model(max_grad=50.0, **cfg); batch = synth.make_batch(**cfg)
    feats = CuMatrix.from_numpy(batch.feats); diff = CuMatrix(batch.T * batch.S, cfg["K"])
    net = Net.from_layers(layers); net.SetTrainOptions(4e-5, 0.9)
    def step():
        net.SetSeqLengths(batch.lens); o = net.Propagate(feats)
        ctc.EvalParallel(batch.lens, o, batch.labels, diff, want_pzx=False)
        net.Backpropagate(diff)
    for _ in range(3): step()
    net.Synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    net.Synchronize(); dt = (time.perf_counter() - t0) / n
    net.SetProfiling(True, accumulate=True)
    for _ in range(n): step()
    net.Synchronize()
    ph = {k: round(1e3 * v / n, 2) for k, v in net.PhaseTimes().items()}
    print(f"{tag:34s} {1e3 * dt:7.2f} ms  {ph}", flush=True)
    if keep: return net, feats, diff
    del net, feats, diff; gc.collect()

R10 = dict(H=320, D=120, S=10, T=1000)
def r10(tag, keep=False): return timed(tag, "cfg2", R10, keep=keep)
def churn(n=200, seed=1):
    rng = np.random.default_rng(seed)
    held = []
    for i in range(n):
        rows = int(rng.integers(1, 4000)); cols = int(rng.choice([4, 40, 320, 1024, 4096]))
        held.append(CuMatrix(rows, cols, zero=False))
        if len(held) > 20 and rng.random() < 0.6: held.pop(int(rng.integers(0, len(held))))
    del held; gc.collect()

mode = sys.argv[1] if len(sys.argv) > 1 else "a"
if mode == "a":      # a freed large Net in front
    r10("first"); timed("cfg4 (3+20 steps, then freed)", "cfg4", {}, n=5); r10("after cfg4"); r10("again")
elif mode == "b":    # churn in front of the first Net
    churn(); r10("after churn"); churn(400, 2); r10("after more churn")
elif mode == "c":    # the same Net shape, again and again
    for i in range(4): r10(f"net {i}")
elif mode == "d":    # a second Net alive beside it
    k = r10("first (kept alive)", keep=True); r10("second beside the first"); del k; gc.collect(); r10("third, first freed")
elif mode == "e":    # a freed cfg2 / cfg5-sized Net in front
    r10("first"); timed("cfg2 S=64", "cfg2", dict(S=64), n=5); r10("after cfg2 S=64"); timed("cfg5 T=1000", "cfg5", dict(T=1000), n=3); r10("after cfg5")
