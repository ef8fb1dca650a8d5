cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_r1g $O/pmc4_*
rocprofv3 --kernel-trace -d $O/prof_r1g -o r1g -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_r1g.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc4_$c -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc4_$c.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/pmc4_SQ -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc4_SQ.log 2>&1
python $R/bench.py --steps 10 --warmup 3 > $O/bench_r1g.json 2> $O/bench_r1g.err
ls -la $O/prof_r1g $O/pmc4_*; tail -c 600 $O/bench_r1g.json
