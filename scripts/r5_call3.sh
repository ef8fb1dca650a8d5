#!/bin/bash
# Round 5, GPU call 3: the two-tile backward kernel (lstm_bwd_persistent_q4_kernel<., 8>) and the per-minibatch overlap rule at
# --num-sequence 64 (cfg2 shape and the recipe shape), then the WHOLE GPU suite and the bench line on this tree.
mkdir -p gpurun_out/r5c; O=gpurun_out/r5c
export TMPDIR=/tmp EESEN_PARITY_OUT=$PWD/$O
( timeout 300 python -m pytest -x -q tests/test_gpu_parity.py -k "two_sequence_tiles or backward_tiles or ctc_multi" 2>&1 | tail -8 ) > $O/tests_new.log 2>&1; cat $O/tests_new.log
one() { local label=$1; shift
  ( timeout 150 env "$@" python bench.py --main-only --S 64 --steps 6 --warmup 2 2>$O/s64_$label.err | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); p=d.get('phase_ms_per_step',{})
        print('$label', round(d['ms_per_step'],2), 'ms', round(d['value']), 'fps', {k: round(v,2) for k,v in p.items() if not k.startswith('ctc')}, flush=True)" ) >> $O/s64.log 2>&1
  grep -a -h "WARNING\|recover" $O/s64_$label.err | head -3 >> $O/s64.log; }
one default A=1
one overlap1 EESEN_OVERLAP=1
one st8off EESEN_BWD_Q4_ST8=0
one st8off_overlap1 EESEN_BWD_Q4_ST8=0 EESEN_OVERLAP=1
one narrow2off EESEN_FWD_NARROW2=0
one narrow2off_st8off EESEN_FWD_NARROW2=0 EESEN_BWD_Q4_ST8=0
cat $O/s64.log
rec() { local label=$1; shift
  ( timeout 200 env "$@" python -c "
import json, bench
for S in (32, 64):
    r = bench.recipe_leg(0, S, 256, 100000)
    print('$label recipe S', S, round(r['ms_per_minibatch'], 2), 'ms/minibatch', round(r['padded_frames_per_s']), 'padded fps', r['persistent_layer_passes'], flush=True)
" 2>/dev/null ) >> $O/recipe.log 2>&1; }
rec default A=1
rec overlap1 EESEN_OVERLAP=1
rec overlap0 EESEN_OVERLAP=0
cat $O/recipe.log
( timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/test_gpu.log 2>&1; cat $O/test_gpu.log
( timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_stderr.log | tail -1 ) > $O/bench_line.json
python - <<'P'
import json
try:
    d = json.load(open("gpurun_out/r5c/bench_line.json"))
    c = d["config"]
    print("bench", round(d["ms_per_step"], 2), "ms", round(d["value"]), d["unit"], "resident", round(c["device_resident_ms_per_step"], 2), "frac", round(d["roofline"]["frac"], 4))
    for k, v in c.get("secondary", {}).items():
        print(" ", k, {a: (round(b, 2) if isinstance(b, float) else b) for a, b in v.items() if a in ("ms_per_step", "ms_per_minibatch", "frames_per_s", "padded_frames_per_s", "error")})
        if "ctc" in v:
            t = v["ctc"]; print("     ctc", round(t["ms"], 3), "ms", round(t["achieved"]), "GB/s; sweep us/step", round(t["sweep"]["us_per_lattice_step"], 3))
except Exception as e:
    print("bench line unreadable:", e)
P
