#!/bin/bash
# Round 6, call 9: the rewritten bounds kernels; the whole GPU suite; the cfg2 / cfg4 / cfg5 steps; the cfg2 timeline.
mkdir -p gpurun_out/r6i; O=gpurun_out/r6i
export TMPDIR=/tmp EESEN_PARITY_OUT=$PWD/$O
run() {
  local label=$1; shift
  ( timeout 400 python bench.py --main-only --steps 10 --warmup 3 "$@" 2>$O/$label.err | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); p=d.get('phase_ms_per_step',{})
        print('$label', 'step', round(d['ms_per_step'],2), 'ms;', {k: round(v,2) for k,v in p.items()}, flush=True)" ) >> $O/ab.log 2>&1
}
run cfg2; run cfg2
EESEN_GEMM_MODE=split EESEN_FWD_F16=0 run cfg2_round5_arithmetic
run cfg4 --config cfg4
run cfg4_bf16fwd --config cfg4 --forward-precision bf16
run cfg5 --config cfg5 --steps 3 --warmup 1
run cfg2_S64 --S 64
cat $O/ab.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --main-only > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_summary.py $(find $O/prof -name "*_results.db" | head -1) > $O/kernel_stats.md
python scripts/timeline.py $(find $O/prof -name "*_results.db" | head -1) > $O/step_timeline.txt 2>/dev/null
rm -rf $O/prof
head -24 $O/kernel_stats.md
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $O/test_gpu.log 2>&1; cat $O/test_gpu.log
