#!/bin/bash
# first-poll delays on the other shapes
mkdir -p gpurun_out/r6q; O=gpurun_out/r6q
export TMPDIR=/tmp
: > $O/poll_sweep3.log
for leg in cfg4 cfg2_S64 wsj_recipe_shape_S10 wsj_recipe_shape_S32; do for f in 300 400 500; do for b in 280 420 560; do
  ( EESEN_POLL_NS=$f,$b timeout 200 python bench.py --leg $leg 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('$leg fwd $f bwd $b', round(d.get('ms_per_step', d.get('ms_per_minibatch', 0)),2), 'ms')" ) >> $O/poll_sweep3.log
done; done; done
cat $O/poll_sweep3.log
