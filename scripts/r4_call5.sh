#!/bin/bash
# round 4, GPU call 5: the forward recurrence on the bf16 pipe -- fp32-class 3-way split at cfg2 (parity, A/B), bf16 forward with W_m as hi + lo
mkdir -p gpurun_out/r4e; O=gpurun_out/r4e
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_bf16_forward.py -q -x 2>&1 | tail -8 ) > $O/test_bf16.log 2>&1
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_dropout.py -q 2>&1 | tail -25 ) > $O/test_split.log 2>&1
one() { local label=$1; shift
  ( env "$@" 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); p=d.get('phase_ms_per_step',{})
        print('$label', round(d['ms_per_step'],2), 'fwd', round(p.get('recurrence_fwd',0),2), 'bwd', round(p.get('recurrence_bwd',0),2), 'in_gemm', round(p.get('input_gemm',0),2), flush=True)" ) >> $O/ab.log 2>&1; }
for round in 1 2; do
  one cfg2_f32rec      EESEN_FWD_SPLIT=0 python bench.py --main-only --steps 10 --warmup 3
  one cfg2_split       EESEN_FWD_SPLIT=1 python bench.py --main-only --steps 10 --warmup 3
  one cfg2_f32rec_noov EESEN_FWD_SPLIT=0 EESEN_OVERLAP=0 python bench.py --main-only --steps 10 --warmup 3
  one cfg2_split_noov  EESEN_FWD_SPLIT=1 EESEN_OVERLAP=0 python bench.py --main-only --steps 10 --warmup 3
  one cfg4_bf16        python bench.py --config cfg4 --main-only --steps 5 --warmup 2 --forward-precision bf16
done
( EESEN_TRACE=1 EESEN_FWD_SPLIT=1 EESEN_OVERLAP=0 python bench.py --main-only --steps 2 --warmup 1 2>&1 | grep EESEN_TRACE ) > $O/trace.log 2>&1
cat $O/test_bf16.log $O/test_split.log $O/ab.log $O/trace.log
