#!/bin/bash
# round-3 profile collection: cfg2 (kernel trace, PMC passes in separate runs, timeline, bench line) + cfg4 kernel trace / timeline
TAG=${1:-r03}; bash scripts/collect_profiles.sh $TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_${TAG}_cfg4
rocprofv3 --kernel-trace -d $O/prof_${TAG}_cfg4 -o ${TAG}cfg4 -- python $R/bench.py --config cfg4 --steps 2 --warmup 1 --main-only > $O/prof_${TAG}_cfg4.log 2>&1
export EESEN_FWD_MID=0   # (counter passes: see collect_profiles.sh)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/pmc_${TAG}_cfg4_SQ -o pmc -- python $R/bench.py --config cfg4 --steps 1 --warmup 1 --main-only > $O/pmc_${TAG}_cfg4_SQ.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${TAG}_cfg4_$c -o pmc -- python $R/bench.py --config cfg4 --steps 1 --warmup 1 --main-only > $O/pmc_${TAG}_cfg4_$c.log 2>&1
done
cd $R
python scripts/rocpd_summary.py $(find $O/prof_${TAG}_cfg4 -name "*_results.db" | head -1) > $O/${TAG}_cfg4_kernel_stats.md
python scripts/timeline.py $(find $O/prof_${TAG}_cfg4 -name "*_results.db" | head -1) > $O/${TAG}_cfg4_step_timeline.txt 2>/dev/null
python scripts/rocpd_pmc_summary.py $(find $O/pmc_${TAG}_cfg4_SQ -name "*.db") > $O/${TAG}_cfg4_pmc_sq.md
python scripts/rocpd_pmc_summary.py $(find $O/pmc_${TAG}_cfg4_FETCH_SIZE $O/pmc_${TAG}_cfg4_WRITE_SIZE -name "*.db") > $O/${TAG}_cfg4_pmc_fetch_write.md
head -12 $O/${TAG}_cfg4_kernel_stats.md; head -8 $O/${TAG}_cfg4_pmc_sq.md
