#!/bin/bash
# same-box A/B: the tuned first-poll delays (default) against what the flight-derived ones were (near-mode reading 390,270; far-mode 600,420)
mkdir -p gpurun_out/r6r; O=gpurun_out/r6r
export TMPDIR=/tmp
: > $O/ab.log
hl() { ( env "$@" timeout 120 python bench.py --main-only --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('headline [$*]', round(d['ms_per_step'],2), 'ms')" ) >> $O/ab.log; }
leg() { local l=$1; shift; ( env "$@" timeout 300 python bench.py --leg $l 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('$l [$*]', round(d.get('ms_per_step', d.get('ms_per_minibatch', 0)),2), 'ms')" ) >> $O/ab.log; }
for i in 1 2 3; do hl X=1; hl EESEN_POLL_NS=390,270; hl EESEN_POLL_NS=600,420; done
for i in 1 2; do leg cfg2_S64 X=1; leg cfg2_S64 EESEN_POLL_NS=390,270; leg cfg2_S64 EESEN_POLL_NS=600,420; done
for i in 1 2; do leg cfg4 X=1; leg cfg4 EESEN_POLL_NS=390,270; leg cfg4 EESEN_POLL_NS=600,420; done
for i in 1 2; do leg wsj_recipe_shape_S32 X=1; leg wsj_recipe_shape_S32 EESEN_POLL_NS=390,270; leg wsj_recipe_shape_S32 EESEN_POLL_NS=600,420; done
sort $O/ab.log
