#!/usr/bin/env python3
"""The feature front end alone (bench.py's `frontend` leg), for `rocprofv3 --kernel-trace --stats`."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.frontend_leg(0, iters=50)))
