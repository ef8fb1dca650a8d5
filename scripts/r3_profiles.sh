#!/bin/bash
# round-3 profile collection: cfg2 (kernel trace, PMC passes in separate runs, timeline, bench line) + cfg4 kernel trace / timeline
bash scripts/collect_profiles.sh r03
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_r03_cfg4
rocprofv3 --kernel-trace -d $O/prof_r03_cfg4 -o r03cfg4 -- python $R/bench.py --config cfg4 --steps 2 --warmup 1 --main-only > $O/prof_r03_cfg4.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/pmc_r03_cfg4_SQ -o pmc -- python $R/bench.py --config cfg4 --steps 1 --warmup 1 --main-only > $O/pmc_r03_cfg4_SQ.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_r03_cfg4_$c -o pmc -- python $R/bench.py --config cfg4 --steps 1 --warmup 1 --main-only > $O/pmc_r03_cfg4_$c.log 2>&1
done
cd $R
python scripts/rocpd_summary.py $(find $O/prof_r03_cfg4 -name "*_results.db" | head -1) > $O/r03_cfg4_kernel_stats.md
python scripts/timeline.py $(find $O/prof_r03_cfg4 -name "*_results.db" | head -1) > $O/r03_cfg4_step_timeline.txt 2>/dev/null
python scripts/rocpd_pmc_summary.py $(find $O/pmc_r03_cfg4_SQ -name "*.db") > $O/r03_cfg4_pmc_sq.md
python scripts/rocpd_pmc_summary.py $(find $O/pmc_r03_cfg4_FETCH_SIZE $O/pmc_r03_cfg4_WRITE_SIZE -name "*.db") > $O/r03_cfg4_pmc_fetch_write.md
head -12 $O/r03_cfg4_kernel_stats.md; head -8 $O/r03_cfg4_pmc_sq.md
