#!/bin/bash
# Round 5, closing GPU call -- everything the committed records of the FINAL tree come from: the whole GPU suite (full-size parity records
# into $O/parity_fullsize.json), smoke(), the four rocprofv3 passes + the driver's bench line (scripts/collect_profiles.sh r05), the
# headline three more times (spread), soaks of the persistent kernels at --num-sequence 32 and 64, and the kernel timeline of one
# --num-sequence 64 step.
mkdir -p gpurun_out/r5z; O=gpurun_out/r5z
export TMPDIR=/tmp EESEN_PARITY_OUT=$PWD/$O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/test_gpu.log 2>&1; cat $O/test_gpu.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $O/smoke.log; cat $O/smoke.log
bash scripts/collect_profiles.sh r05 > $O/collect.log 2>&1; tail -45 $O/collect.log
for i in 1 2 3; do
  ( timeout 120 python bench.py --main-only --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('headline run $i', round(d['ms_per_step'],3), 'ms', round(d['value']), 'fps', flush=True)" ) >> $O/headline_spread.log 2>&1
done
cat $O/headline_spread.log
( timeout 200 python scripts/soak.py cfg2 300 2>&1 | tail -1; timeout 200 python scripts/soak.py cfg2 300 64 2>&1 | tail -1 ) > $O/soak.log 2>&1; cat $O/soak.log
R=$GRAFT_REPO_ROOT
cd /tmp && ( timeout 200 rocprofv3 --kernel-trace -d $R/$O/prof -o s64 -- python $R/bench.py --steps 3 --warmup 1 --main-only --S 64 > $R/$O/prof.log 2>&1 )
cd $R
DB=$(find $O/prof -name "*_results.db" | head -1)
if [ -n "$DB" ]; then
  python scripts/rocpd_summary.py $DB > $O/r05_s64_kernel_stats.md
  python scripts/timeline.py $DB > $O/r05_s64_step_timeline.txt 2>/dev/null
  head -12 $O/r05_s64_kernel_stats.md; tail -3 $O/r05_s64_step_timeline.txt
  rm -rf $O/prof
fi
rm -rf gpurun_out/prof_r05 gpurun_out/pmc_r05_*     # the rocpd databases: summarised above, too large to travel back
