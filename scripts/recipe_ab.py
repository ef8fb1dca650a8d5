#!/usr/bin/env python3
"""The recipe-shape leg of bench.py (4 x 320, variable lengths through the trainer's path) at one --num-sequence, for A/B runs under
different environment switches:   EESEN_GEMM_MODE=split python scripts/recipe_ab.py 10"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 10
r = bench.recipe_leg(0, ns) if ns <= 20 else bench.recipe_leg(0, ns, 256, 100000)
print(json.dumps({k: r[k] for k in ("ms_per_minibatch", "ms_per_minibatch_min_median_max", "padded_frames_per_s", "minibatches", "persistent_layer_passes", "recoveries")}))
