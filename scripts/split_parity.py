#!/usr/bin/env python3
"""profiles/parity_cfg{2,3,4,5}.json from the record the full-size parity tests write (tests/test_gpu_reference_fullsize.py ->
$EESEN_PARITY_OUT/parity_fullsize.json):  scripts/split_parity.py gpurun_out/r4k/parity_fullsize.json [COMMIT]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
reps = json.load(open(sys.argv[1]))
commit = sys.argv[2] if len(sys.argv) > 2 else ""
sys.path.insert(0, ROOT)
from eesen_amd.build import csrc_digest  # noqa: E402
groups = {"cfg2": [], "cfg3": [], "cfg4": [], "cfg5": [], "recipe320": []}
for r in reps:
    for k in groups:
        if ("full_" + k) in r["case"]:
            groups[k].append(r)
for k, rs in groups.items():
    if rs:
        out = dict(source="tests/test_gpu_reference_fullsize.py on an MI355X (pytest -m gpu); one record per test case", commit=commit, csrc_sha=csrc_digest(), records=rs)
        json.dump(out, open(os.path.join(ROOT, "profiles", f"parity_{k}.json"), "w"), indent=1)
        print(k, [r["case"] + ("" if r.get("persistent", True) else " (per-step kernels)") for r in rs])
