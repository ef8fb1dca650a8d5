# round-2 profile collection: kernel trace (+ stats summary, step timeline), PMC passes (separate runs), bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r02b}
rm -rf $O/prof_$TAG $O/pmc_${TAG}_*
timeout 600 rocprofv3 --kernel-trace -d $O/prof_$TAG -o $TAG -- python $R/bench.py --steps 3 --warmup 1 --main-only > $O/prof_$TAG.log 2>&1
# counter passes let one kernel run at a time: the side stream's milestone waiter could be picked before the recurrence it waits for
# (it then runs into its wall-clock bound: one warning, one lost minibatch, no early GEMM from there on).  The recurrence kernels are
# the same with EESEN_FWD_MID=0; the input GEMM is one launch instead of three.  (A command-processor wait -- hipStreamWaitValue64 --
# was tried as the arm for these passes in round 4: it dead-locked under --pmc for the whole 40 minutes of the call.)
export EESEN_FWD_MID=0
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${TAG}_$c -o pmc -- python $R/bench.py --steps 2 --warmup 1 --main-only > $O/pmc_${TAG}_$c.log 2>&1
done
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $O/pmc_${TAG}_SQ -o pmc -- python $R/bench.py --steps 2 --warmup 1 --main-only > $O/pmc_${TAG}_SQ.log 2>&1
GEMM_BENCH_ONLY="one tile" timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_${TAG}_calib -o pmc -- python $R/scripts/gemm_bench.py > $O/pmc_${TAG}_calib.log 2>&1
unset EESEN_FWD_MID
cd $R
python scripts/rocpd_pmc_summary.py $(find $O/pmc_${TAG}_calib -name "*.db") > $O/${TAG}_pmc_fetch_calibration.md
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_$TAG.json 2> $O/bench_$TAG.err
python scripts/rocpd_summary.py $(find $O/prof_$TAG -name "*_results.db" | head -1) > $O/${TAG}_kernel_stats.md
python scripts/rocpd_pmc_summary.py $(find $O/pmc_${TAG}_FETCH_SIZE $O/pmc_${TAG}_WRITE_SIZE -name "*.db") > $O/${TAG}_pmc_fetch_write.md
python scripts/rocpd_pmc_summary.py $(find $O/pmc_${TAG}_SQ -name "*.db") > $O/${TAG}_pmc_sq.md
python scripts/timeline.py $(find $O/prof_$TAG -name "*_results.db" | head -1) > $O/${TAG}_step_timeline.txt 2>/dev/null
head -25 $O/${TAG}_kernel_stats.md; head -14 $O/${TAG}_pmc_fetch_write.md; tail -c 400 $O/bench_$TAG.json
