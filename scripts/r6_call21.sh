#!/bin/bash
mkdir -p gpurun_out/r6q; O=gpurun_out/r6q
export TMPDIR=/tmp
: > $O/poll_sweep4.log
one() { # leg f b
  ( EESEN_POLL_NS=$2,$3 timeout 300 python bench.py --leg $1 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('$1 fwd $2 bwd $3', round(d.get('ms_per_step', d.get('ms_per_minibatch', 0)),2), 'ms')" ) >> $O/poll_sweep4.log
}
for f in 500 600 700 850; do for b in 140 200 280; do one cfg2_S64 $f $b; done; done
for f in 150 200 300; do for b in 140 200 280; do one cfg4 $f $b; done; done
for f in 200 300 400; do for b in 140 200 280; do one wsj_recipe_shape_S10 $f $b; one wsj_recipe_shape_S32 $f $b; done; done
for f in 200 300 400; do for b in 200 280 420; do one cfg5 $f $b; done; done
cat $O/poll_sweep4.log
