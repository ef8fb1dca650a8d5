#!/bin/bash
# Round 6, call 5: the two-plane fp16 GEMM (mode 2): accuracy against fp64 and the denormal check, the GEMM microbenchmark in the
# three modes, and the cfg2 / cfg4 / cfg5 steps with EESEN_GEMM_MODE=half against the default on the same box.
mkdir -p gpurun_out/r6e; O=gpurun_out/r6e
export TMPDIR=/tmp EESEN_PARITY_OUT=$PWD/$O
( timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q 2>&1 | tail -15 ) > $O/test_gemm.log 2>&1; cat $O/test_gemm.log
( timeout 300 python scripts/gemm_bench.py 2>&1 ) > $O/gemm_bench.log; cat $O/gemm_bench.log
run() {  # label, config args..., env via EXTRA
  local label=$1; shift
  ( timeout 400 python bench.py --main-only --steps 10 --warmup 3 "$@" 2>$O/$label.err | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); p=d.get('phase_ms_per_step',{})
        print('$label', 'step', round(d['ms_per_step'],2), 'ms;', {k: round(v,2) for k,v in p.items()}, flush=True)" ) >> $O/ab.log 2>&1
}
for r in 1 2; do
  EESEN_GEMM_MODE=split run cfg2_split
  EESEN_GEMM_MODE=half run cfg2_half
done
EESEN_GEMM_MODE=split run cfg4_split --config cfg4
EESEN_GEMM_MODE=half run cfg4_half --config cfg4
EESEN_GEMM_MODE=split run cfg4bf_split --config cfg4 --forward-precision bf16
EESEN_GEMM_MODE=half run cfg4bf_half --config cfg4 --forward-precision bf16
EESEN_GEMM_MODE=split run cfg5_split --config cfg5 --steps 3 --warmup 1
EESEN_GEMM_MODE=half run cfg5_half --config cfg5 --steps 3 --warmup 1
cat $O/ab.log
( EESEN_GEMM_MODE=half timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_fullsize.py -x -q 2>&1 | tail -15 ) > $O/test_parity_half.log 2>&1; cat $O/test_parity_half.log
