#!/bin/bash
# Round 5, GPU call 7: the 16 x 8 bf16-pipe forward tile at <= 16 sequences (recipes' default --num-sequence 10), A/B on one box.
mkdir -p gpurun_out/r5g; O=gpurun_out/r5g
export TMPDIR=/tmp
rec() { local label=$1; shift
  ( timeout 200 env "$@" python -c "
import json, bench
for S, n, lim in ((10, 120, 25000), (16, 120, 25000)):
    r = bench.recipe_leg(0, S, n, lim)
    print('$label recipe S', S, round(r['ms_per_minibatch'], 2), 'ms/minibatch', round(r['padded_frames_per_s']), 'padded fps', r['persistent_layer_passes'], flush=True)
" 2>/dev/null ) >> $O/recipe.log 2>&1; }
rec default A=1
rec t16small EESEN_FWD_T16_SMALL=1
rec default_again A=1
rec t16small_again EESEN_FWD_T16_SMALL=1
cat $O/recipe.log
( EESEN_FWD_T16_SMALL=1 timeout 300 python -m pytest -x -q tests/test_gpu_parity.py -k "recipe_shape or odd_shapes or unaligned or train_step_parity or persistent_recurrence_matches" 2>&1 | tail -5 ) > $O/tests_t16.log 2>&1; cat $O/tests_t16.log
