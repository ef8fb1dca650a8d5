#!/usr/bin/env python3
"""Does the recipe-shape leg at --num-sequence 10 depend on what ran in the process before it?  (bench.py's line: 23.3 ms per minibatch after the
cfg legs; alone: 18.9.)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
def r10(tag):
    r = bench.recipe_leg(0, 10)
    print(tag, round(r["ms_per_minibatch"], 2), r["persistent_layer_passes"], flush=True)
r10("first")
for name in sys.argv[1:]:
    s = bench.secondary_leg(name, 0, steps=2, warmup=1, reps=1, over=(dict(S=64) if name == "cfg2" else None))
    print(name, round(s["ms_per_step"], 2), flush=True)
    r10("after " + name)
