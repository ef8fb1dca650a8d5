#!/bin/bash
# Round 5, GPU call 6: ragged sequence tiles on the 4 x 32 backward kernel (the recipes' default --num-sequence 10), A/B on one box.
mkdir -p gpurun_out/r5f; O=gpurun_out/r5f
export TMPDIR=/tmp
( timeout 400 python -m pytest -x -q tests/test_gpu_parity.py -k "backward_tiles or two_sequence_tiles or recipe_shape or odd_shapes or unaligned or train_step_parity or cell_counts" 2>&1 | tail -8 ) > $O/tests_new.log 2>&1; cat $O/tests_new.log
rec() { local label=$1; shift
  ( timeout 200 env "$@" python -c "
import json, bench
for S, n, lim in ((10, 120, 25000), (20, 120, 25000)):
    r = bench.recipe_leg(0, S, n, lim)
    print('$label recipe S', S, round(r['ms_per_minibatch'], 2), 'ms/minibatch', round(r['padded_frames_per_s']), 'padded fps', r['persistent_layer_passes'], flush=True)
" 2>/dev/null ) >> $O/recipe.log 2>&1; }
rec default A=1
rec q4off EESEN_BWD_Q4=0
cat $O/recipe.log
