#!/bin/bash
# round-4 profile collection: cfg2 (kernel trace; PMC passes in separate runs with EESEN_FWD_MID=0; timeline; the default bench line) + cfg4 fp32 and cfg4 bf16-forward kernel traces / timelines
TAG=${1:-r04}; bash scripts/collect_profiles.sh $TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for v in f32 bf16; do
  rm -rf $O/prof_${TAG}_cfg4_$v
  timeout 400 rocprofv3 --kernel-trace -d $O/prof_${TAG}_cfg4_$v -o ${TAG}cfg4$v -- python $R/bench.py --config cfg4 --steps 2 --warmup 1 --main-only --forward-precision $v > $O/prof_${TAG}_cfg4_$v.log 2>&1
done
cd $R
for v in f32 bf16; do
  python scripts/rocpd_summary.py $(find $O/prof_${TAG}_cfg4_$v -name "*_results.db" | head -1) > $O/${TAG}_cfg4_${v}_kernel_stats.md
  python scripts/timeline.py $(find $O/prof_${TAG}_cfg4_$v -name "*_results.db" | head -1) > $O/${TAG}_cfg4_${v}_step_timeline.txt 2>/dev/null
done
timeout 300 python bench.py --config cfg4 --steps 5 --warmup 2 --main-only > $O/bench_${TAG}_cfg4_f32.json 2>/dev/null
timeout 300 python bench.py --config cfg4 --steps 5 --warmup 2 --main-only --forward-precision bf16 > $O/bench_${TAG}_cfg4_bf16.json 2>/dev/null
head -14 $O/${TAG}_cfg4_bf16_kernel_stats.md; head -10 $O/${TAG}_pmc_sq.md
