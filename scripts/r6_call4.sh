#!/bin/bash
# Round 6, call 4: the plans of every BASELINE shape as data (DESIGN.md's front page is rendered from them), the whole GPU suite on the
# current tree, smoke, one default bench line.
mkdir -p gpurun_out/r6d; O=gpurun_out/r6d
export TMPDIR=/tmp EESEN_PARITY_OUT=$PWD/$O
( timeout 600 python scripts/plans.py > $O/r06_plans.json 2> $O/plans.err ); tail -3 $O/plans.err; python -c "
import json; d=json.load(open('$O/r06_plans.json'))
for k,v in d['shapes'].items(): print(k, round(v['ms_per_step_device_resident'],2), 'ms', v['plan']['layers'][-1]['forward']['kernel'], v['plan']['layers'][-1]['backward']['kernel'], v['us_per_recurrence_step'])"
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/test_gpu.log 2>&1; cat $O/test_gpu.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $O/smoke.log; cat $O/smoke.log
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err ); tail -c 1500 $O/bench_line.json
