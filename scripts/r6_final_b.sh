#!/bin/bash
# Round 6, closing GPU calls, part B: counters and timelines of the wide configurations on the final tree (scripts/collect_profiles_wide.sh
# r06), soaks of the persistent kernels at --num-sequence 32 and 64, the kernel timeline of one --num-sequence 64 step.
mkdir -p gpurun_out/r6z; O=gpurun_out/r6z
export TMPDIR=/tmp
bash scripts/collect_profiles_wide.sh r06 > $O/collect_wide.log 2>&1; tail -60 $O/collect_wide.log
( timeout 200 python scripts/soak.py cfg2 300 2>&1 | tail -1; timeout 200 python scripts/soak.py cfg2 300 64 2>&1 | tail -1 ) > $O/soak.log 2>&1; cat $O/soak.log
R=$GRAFT_REPO_ROOT
cd /tmp && ( timeout 200 rocprofv3 --kernel-trace -d $R/$O/prof -o s64 -- python $R/bench.py --steps 3 --warmup 1 --main-only --S 64 > $R/$O/prof.log 2>&1 )
cd $R
DB=$(find $O/prof -name "*_results.db" | head -1)
if [ -n "$DB" ]; then
  python scripts/rocpd_summary.py $DB > $O/r06_s64_kernel_stats.md
  python scripts/timeline.py $DB > $O/r06_s64_step_timeline.txt 2>/dev/null
  head -12 $O/r06_s64_kernel_stats.md; tail -3 $O/r06_s64_step_timeline.txt
  rm -rf $O/prof
fi
