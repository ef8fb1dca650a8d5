#!/bin/bash
# first-poll delays of the hand-off waits, swept on the final kernels (the factors date from round 2's kernels)
mkdir -p gpurun_out/r6q; O=gpurun_out/r6q
export TMPDIR=/tmp
: > $O/poll_sweep.log
for f in 0 100 200 300 400 600; do for b in 0 100 200 280 420; do
  ( EESEN_POLL_NS=$f,$b timeout 120 python bench.py --main-only --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('fwd $f bwd $b', round(d['ms_per_step'],2), 'ms', {k: round(v,2) for k,v in d['config'].get('phases_ms',{}).items() if 'recurrence' in k})" ) >> $O/poll_sweep.log
done; done
cat $O/poll_sweep.log
