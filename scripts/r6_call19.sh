#!/bin/bash
mkdir -p gpurun_out/r6q; O=gpurun_out/r6q
export TMPDIR=/tmp
: > $O/poll_sweep2.log
for rep in 1 2; do for f in 350 400 450 500; do for b in 420 500 600 700 850 1000 1300; do
  ( EESEN_POLL_NS=$f,$b timeout 120 python bench.py --main-only --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('rep $rep fwd $f bwd $b', round(d['ms_per_step'],2), 'ms')" ) >> $O/poll_sweep2.log
done; done; done
python - <<'PY'
import re,collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r6q/poll_sweep2.log'):
    m=re.match(r'rep \d+ fwd (\d+) bwd (\d+) ([\d.]+)',l)
    if m: d[(int(m[1]),int(m[2]))].append(float(m[3]))
for k in sorted(d): print(k, d[k])
PY
