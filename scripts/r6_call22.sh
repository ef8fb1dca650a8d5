#!/bin/bash
mkdir -p gpurun_out/r6r; O=gpurun_out/r6r
export TMPDIR=/tmp
: > $O/after.log
for i in 1 2 3; do ( EESEN_PRINT_FLIGHT=1 timeout 120 python bench.py --main-only --steps 20 --warmup 5 2>$O/err.txt | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('headline run $i', round(d['ms_per_step'],2), 'ms', end=' ')"; grep -h "increment flight" $O/err.txt | head -1 ) >> $O/after.log; done
for leg in cfg2_S64 cfg4 cfg4_bf16_forward wsj_recipe_shape_S10 wsj_recipe_shape_S20 wsj_recipe_shape_S32 wsj_recipe_shape_S64 cfg5; do
  ( timeout 300 python bench.py --leg $leg 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('$leg', round(d.get('ms_per_step', d.get('ms_per_minibatch', 0)),2), 'ms')" ) >> $O/after.log
done
cat $O/after.log
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -m gpu -q -x 2>&1 | tail -4 ) > $O/tests.log; cat $O/tests.log
