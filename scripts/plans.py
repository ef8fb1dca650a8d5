#!/usr/bin/env python3
"""What the library's selection functions choose for every BASELINE configuration and the recipe shapes, as data:
   python scripts/plans.py > profiles/r06_plans.json        (needs a GPU: the plans ask the device for occupancy and registers)
Per shape: eesen_net_plan_string (the plans the launchers execute), and one timed training step's per-phase HIP-event times.
tools/front_page.py renders DESIGN.md's front page from this file."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eesen_amd import synth  # noqa: E402
from eesen_amd.api import Net, Ctc, CuMatrix  # noqa: E402
from eesen_amd.build import csrc_digest  # noqa: E402

SHAPES = [
    ("cfg2", "cfg2", {}, 0, "BASELINE configs[1]: the headline (4 x 512, S = 32, T = 1000)"),
    ("cfg2_S64", "cfg2", dict(S=64), 0, "the headline net at --num-sequence 64"),
    ("recipe320_S10", "cfg2", dict(H=320, D=120, S=10), 0, "the recipes' own width (4 x 320 on 120-d features) at their default --num-sequence 10"),
    ("recipe320_S32", "cfg2", dict(H=320, D=120, S=32), 0, "the recipes' width at --num-sequence 32"),
    ("cfg4", "cfg4", {}, 0, "BASELINE configs[3] in fp32 (5 x 1024 + 512-d projections, S = 32, T = 1000)"),
    ("cfg4_bf16_forward", "cfg4", {}, 1, "BASELINE configs[3] as quoted: bf16 forward / fp32 CTC + backward"),
    ("cfg5", "cfg5", {}, 0, "BASELINE configs[4]: 6 x 1024, S = 64 per GPU, T = 3000"),
]


def main():
    out = {"csrc_sha": csrc_digest(), "shapes": {}}
    for name, cfgname, over, fwd_bf16, what in SHAPES:
        cfg = synth.config(cfgname); cfg.update(over)
        layers = synth.make_model(max_grad=50.0, **cfg)
        batch = synth.make_batch(**cfg)
        net = Net.from_layers(layers); net.SetTrainOptions(4e-5, 0.9); net.SetForwardPrecision(fwd_bf16)
        ctc = Ctc(); ctc.SetGuard(net)
        feats = CuMatrix.from_numpy(batch.feats)
        diff = CuMatrix(batch.T * batch.S, cfg["K"])

        def step():
            net.SetSeqLengths(batch.lens)
            o = net.Propagate(feats)
            ctc.EvalParallel(batch.lens, o, batch.labels, diff, want_pzx=False)
            ctc.ErrorRateMSeq(batch.lens, o, batch.labels, deferred=True)
            net.Backpropagate(diff)
        step(); step(); net.Synchronize()
        plan = net.Plan()
        n = 3 if name == "cfg5" else 5
        net.SetProfiling(True, accumulate=True)
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        net.Synchronize()
        dt = (time.perf_counter() - t0) / n
        ph = net.PhaseTimes(); net.SetProfiling(False)
        info = net.RecurrenceInfo()
        nl = cfg["layers"]
        out["shapes"][name] = {"what": what, "config": {k: cfg[k] for k in ("layers", "H", "D", "K", "S", "T")} | {"proj": cfg.get("proj", 0), "forward_bf16": bool(fwd_bf16)},
                               "plan": plan, "ms_per_step_device_resident": 1e3 * dt, "phase_ms_per_step": {k: 1e3 * v / n for k, v in ph.items()},
                               "us_per_recurrence_step": {"fwd": 1e6 * ph["recurrence_fwd"] / n / nl / batch.T, "bwd": 1e6 * ph["recurrence_bwd"] / n / nl / batch.T},
                               "persistent_layers": info, "recoveries": net.recoveries}
        del net, ctc, feats, diff
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
