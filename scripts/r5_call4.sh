#!/bin/bash
# Round 5, GPU call 4: the census-backed two-narrow-workgroups-per-CU forward grid as the default at --num-sequence 64, A/B on one box.
mkdir -p gpurun_out/r5d; O=gpurun_out/r5d
export TMPDIR=/tmp
( timeout 300 python -m pytest -x -q tests/test_gpu_parity.py -k "two_narrow or two_sequence_tiles or forward_recurrence_arms or persistent_recurrence_matches" 2>&1 | tail -8 ) > $O/tests_new.log 2>&1; cat $O/tests_new.log
one() { local label=$1; shift
  ( timeout 150 env "$@" python bench.py --main-only --S 64 --steps 6 --warmup 2 2>$O/s64_$label.err | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); p=d.get('phase_ms_per_step',{})
        print('$label', round(d['ms_per_step'],2), 'ms', round(d['value']), 'fps', {k: round(v,2) for k,v in p.items() if not k.startswith('ctc')}, flush=True)" ) >> $O/s64.log 2>&1
  grep -a -h "WARNING\|recover" $O/s64_$label.err | head -3 >> $O/s64.log; }
one default A=1
one narrow2off EESEN_FWD_NARROW2=0
one overlap1 EESEN_OVERLAP=1
one s32_default A=1 ; sed -i 's/^s32_default/(that was S=64 again; ignore)/' $O/s64.log
( timeout 100 python bench.py --main-only --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('cfg2 S=32 headline on this box', round(d['ms_per_step'],2), 'ms', round(d['value']), 'fps')" ) >> $O/s64.log 2>&1
cat $O/s64.log
rec() { local label=$1; shift
  ( timeout 200 env "$@" python -c "
import json, bench
for S in (32, 64):
    r = bench.recipe_leg(0, S, 256, 100000)
    print('$label recipe S', S, round(r['ms_per_minibatch'], 2), 'ms/minibatch', round(r['padded_frames_per_s']), 'padded fps', r['persistent_layer_passes'], flush=True)
" 2>/dev/null ) >> $O/recipe.log 2>&1; }
rec default A=1
rec narrow2off EESEN_FWD_NARROW2=0
rec overlap0 EESEN_OVERLAP=0
cat $O/recipe.log
