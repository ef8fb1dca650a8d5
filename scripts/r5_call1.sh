#!/bin/bash
# Round 5, GPU call 1: the new / changed tests (CTC lattices above 1024 positions, multi-wave sweep bit-identity, two ranks each holding
# persistent grids, mask accessors at the model's width, the trainers' rspecifier collective), the H2D-inclusive bench line with the
# cfg2_S64 / cfg5-ctc legs, the CTC waves A/B and the co-run probe at cfg4's bf16 forward.
mkdir -p gpurun_out/r5a; O=gpurun_out/r5a
export TMPDIR=/tmp EESEN_PARITY_OUT=$PWD/$O
( timeout 420 python -m pytest -x -q tests/test_gpu_parity.py -k "ctc or forward_recurrence_arms or backward_tiles or persistent_recurrence_matches or error_behaviour" 2>&1 | tail -12 ) > $O/tests_parity.log 2>&1
cat $O/tests_parity.log
( timeout 600 python -m pytest -x -q tests/test_gpu_multirank.py tests/test_gpu_bf16_forward.py tests/test_gpu_dropout.py tests/test_gpu_cli.py::test_shared_list_dealing_is_opt_in tests/test_gpu_parallel.py 2>&1 | tail -12 ) > $O/tests_other.log 2>&1
cat $O/tests_other.log
( timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_stderr.log | tail -1 ) > $O/bench_line.json
python - <<'P'
import json
try:
    d = json.load(open("gpurun_out/r5a/bench_line.json"))
    c = d["config"]
    print("bench", round(d["ms_per_step"], 2), "ms", round(d["value"]), d["unit"], "resident", round(c["device_resident_ms_per_step"], 2), "frac", round(d["roofline"]["frac"], 4))
    for k, v in c.get("secondary", {}).items():
        print(" ", k, {a: (round(b, 2) if isinstance(b, float) else b) for a, b in v.items() if a in ("ms_per_step", "ms_per_minibatch", "frames_per_s", "padded_frames_per_s", "ms_per_step_min_median_max", "persistent_layers", "error")})
        if "phase_ms_per_step" in v:
            print("     phases", {a: round(b, 2) for a, b in v["phase_ms_per_step"].items()})
        if "ctc" in v:
            t = v["ctc"]; print("     ctc", round(t["ms"], 3), "ms", round(t["achieved"]), "GB/s; sweep us/step", round(t["sweep"]["us_per_lattice_step"], 3), "bulk", round(t["bulk"]["ms"], 3), "ms moved GB/s", round(t["bulk"]["moved_GBps"]))
    print("  phases", {a: round(b, 2) for a, b in d["phase_ms_per_step"].items()})
except Exception as e:
    print("bench line unreadable:", e)
P
tail -5 $O/bench_stderr.log
( timeout 200 python scripts/ctc_waves_probe.py 2>$O/ctc_waves.err | tail -1 ) > $O/ctc_waves.json; cat $O/ctc_waves.json; tail -3 $O/ctc_waves.err
( timeout 300 python scripts/corun_probe.py --config cfg4 --forward-bf16 --caps 48,16 2>$O/corun_cfg4.err | tail -1 ) > $O/corun_cfg4_bf16.json; cat $O/corun_cfg4_bf16.json; tail -3 $O/corun_cfg4.err
