#!/bin/bash
# Round 6, closing GPU calls, part A -- the records of the FINAL tree: the whole GPU suite (full-size parity records into
# $O/parity_fullsize.json, GEMM accuracy, multirank soaks), smoke(), the plans of every BASELINE shape (DESIGN.md's front page), the four
# rocprofv3 passes + the driver's bench line at cfg2 (scripts/collect_profiles.sh r06), the headline three more times.
mkdir -p gpurun_out/r6z; O=gpurun_out/r6z
export TMPDIR=/tmp EESEN_PARITY_OUT=$PWD/$O
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $O/test_gpu.log 2>&1; cat $O/test_gpu.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $O/smoke.log; cat $O/smoke.log
( timeout 900 python scripts/plans.py > $O/r06_plans.json 2> $O/plans.err ); tail -3 $O/plans.err; python -c "
import json; d=json.load(open('$O/r06_plans.json'))
for k,v in d['shapes'].items(): print(k, round(v['ms_per_step_device_resident'],2), 'ms', v['plan']['layers'][-1]['forward']['kernel'], v['plan']['layers'][-1]['backward']['kernel'], v['us_per_recurrence_step'])"
bash scripts/collect_profiles.sh r06 > $O/collect.log 2>&1; tail -45 $O/collect.log
for i in 1 2 3; do
  ( timeout 120 python bench.py --main-only --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('headline run $i', round(d['ms_per_step'],3), 'ms', round(d['value']), 'fps', flush=True)" ) >> $O/headline_spread.log 2>&1
done
cat $O/headline_spread.log
rm -rf gpurun_out/prof_r06 gpurun_out/pmc_r06_*     # the rocpd databases: summarised above, too large to travel back
