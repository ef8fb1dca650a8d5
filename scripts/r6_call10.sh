#!/bin/bash
# Round 6, call 10: the K-split backward tile on fp16 planes (EESEN_BWD_F16): parity tests on wide layers, then A/B at cfg4 / cfg5; the
# rewritten bounds pass (tests, cfg2 step).
mkdir -p gpurun_out/r6j; O=gpurun_out/r6j
export TMPDIR=/tmp EESEN_PARITY_OUT=$PWD/$O
( timeout 600 python -m pytest tests/test_gpu_gemm.py -q -k "bounds_pass or accurate or keep_what" 2>&1 | tail -8 ) > $O/test_bounds.log 2>&1; cat $O/test_bounds.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "persistent_recurrence_matches or train_step_parity or backward_tiles" 2>&1 | tail -12 ) > $O/test_parity.log 2>&1; cat $O/test_parity.log
run() {
  local label=$1; shift
  ( timeout 400 python bench.py --main-only --steps 10 --warmup 3 "$@" 2>$O/$label.err | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); p=d.get('phase_ms_per_step',{})
        print('$label', 'step', round(d['ms_per_step'],2), 'ms;', {k: round(v,2) for k,v in p.items()}, d['config']['kernels'] if 'kernels' in d['config'] else '', flush=True)" ) >> $O/ab.log 2>&1
}
run cfg2
EESEN_BWD_F16=0 run cfg4_bwd_f32 --config cfg4
EESEN_BWD_F16=1 run cfg4_bwd_f16 --config cfg4
EESEN_BWD_F16=1 run cfg4bf_bwd_f16 --config cfg4 --forward-precision bf16
EESEN_BWD_F16=0 run cfg5_bwd_f32 --config cfg5 --steps 3 --warmup 1
EESEN_BWD_F16=1 run cfg5_bwd_f16 --config cfg5 --steps 3 --warmup 1
cat $O/ab.log | cut -c1-700
( timeout 1500 python -m pytest tests/test_gpu_reference_fullsize.py tests/test_gpu_fullsize.py -q 2>&1 | tail -12 ) > $O/test_fullsize.log 2>&1; cat $O/test_fullsize.log
