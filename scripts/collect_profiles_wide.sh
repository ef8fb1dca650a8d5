#!/bin/bash
# Final-tree counters for the WIDE configurations: per configuration a kernel trace (stats + step timeline) and
# three PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters) in separate runs -- rocprofv3 --pmc lets one kernel run at a time, so the
# counter passes run with EESEN_FWD_MID=0 (no milestone waiter; the recurrence kernels are the same).
#   scripts/collect_profiles_wide.sh TAG           -> gpurun_out/TAG_{cfg4_f32,cfg4_bf16,cfg5}_{kernel_stats,pmc_fetch_write,pmc_sq}.md, _step_timeline.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r06}
run() {   # name, bench args...
  local name=$1; shift
  rm -rf $O/prof_${TAG}_$name $O/pmc_${TAG}_${name}_*
  timeout 500 rocprofv3 --kernel-trace -d $O/prof_${TAG}_$name -o t -- python $R/bench.py --main-only "$@" > $O/prof_${TAG}_$name.log 2>&1
  export EESEN_FWD_MID=0
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 500 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${TAG}_${name}_$c -o pmc -- python $R/bench.py --main-only "$@" > $O/pmc_${TAG}_${name}_$c.log 2>&1
  done
  timeout 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $O/pmc_${TAG}_${name}_SQ -o pmc -- python $R/bench.py --main-only "$@" > $O/pmc_${TAG}_${name}_SQ.log 2>&1
  unset EESEN_FWD_MID
  ( cd $R
    python scripts/rocpd_summary.py $(find $O/prof_${TAG}_$name -name "*_results.db" | head -1) > $O/${TAG}_${name}_kernel_stats.md
    python scripts/timeline.py $(find $O/prof_${TAG}_$name -name "*_results.db" | head -1) > $O/${TAG}_${name}_step_timeline.txt 2>/dev/null
    python scripts/rocpd_pmc_summary.py $(find $O/pmc_${TAG}_${name}_FETCH_SIZE $O/pmc_${TAG}_${name}_WRITE_SIZE -name "*.db") > $O/${TAG}_${name}_pmc_fetch_write.md
    python scripts/rocpd_pmc_summary.py $(find $O/pmc_${TAG}_${name}_SQ -name "*.db") > $O/${TAG}_${name}_pmc_sq.md
    head -9 $O/${TAG}_${name}_kernel_stats.md; head -8 $O/${TAG}_${name}_pmc_fetch_write.md; head -8 $O/${TAG}_${name}_pmc_sq.md )
  rm -rf $O/prof_${TAG}_$name $O/pmc_${TAG}_${name}_*     # the rocpd databases: summarised above, too large to travel back
}
run cfg4_f32 --config cfg4 --steps 2 --warmup 1
run cfg4_bf16 --config cfg4 --steps 2 --warmup 1 --forward-precision bf16
run cfg5 --config cfg5 --steps 1 --warmup 1
