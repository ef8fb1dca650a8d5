#!/bin/bash
# Round 6, call 7: per-row / per-column operand bounds of the fp16-plane GEMMs (accuracy tests), the tests that failed in call 6,
# schedule A/B in the new regime (side stream on / off, middle-first on / off), the cfg2 timeline.
mkdir -p gpurun_out/r6g; O=gpurun_out/r6g
export TMPDIR=/tmp EESEN_PARITY_OUT=$PWD/$O
( timeout 900 python -m pytest tests/test_gpu_gemm.py "tests/test_gpu_parity.py::test_two_narrow_forward_workgroups_per_cu" tests/test_gpu_fullsize.py::test_cfg5_full_length_layer_persistent_equals_per_step_kernels tests/test_gpu_parity.py::test_persistent_recurrence_matches_step_kernels tests/test_gpu_parity.py::test_train_step_parity -q 2>&1 | tail -30 ) > $O/test_a.log 2>&1; cat $O/test_a.log
( timeout 300 python scripts/gemm_bench.py 2>&1 | grep f16 ) > $O/gemm_bench.log; cat $O/gemm_bench.log
run() {
  local label=$1; shift
  ( timeout 400 python bench.py --main-only --steps 10 --warmup 3 "$@" 2>$O/$label.err | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); p=d.get('phase_ms_per_step',{})
        print('$label', 'step', round(d['ms_per_step'],2), 'ms;', {k: round(v,2) for k,v in p.items()}, flush=True)" ) >> $O/ab.log 2>&1
}
for r in 1 2; do
  run cfg2_default
  EESEN_OVERLAP=0 run cfg2_no_side_stream
  EESEN_FWD_MID=0 run cfg2_no_mid_first
  EESEN_SIDE_LDS_KB=32 run cfg2_side_two_per_cu
done
run cfg4_default --config cfg4
EESEN_OVERLAP=1 run cfg4_side_stream --config cfg4
run cfg2S64_default --S 64
EESEN_OVERLAP=1 run cfg2S64_side_stream --S 64
cat $O/ab.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --main-only > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_summary.py $(find $O/prof -name "*_results.db" | head -1) > $O/kernel_stats.md
python scripts/timeline.py $(find $O/prof -name "*_results.db" | head -1) > $O/step_timeline.txt 2>/dev/null
rm -rf $O/prof
head -30 $O/kernel_stats.md; cat $O/step_timeline.txt | head -80
