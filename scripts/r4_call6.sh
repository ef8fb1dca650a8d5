#!/bin/bash
mkdir -p gpurun_out/r4f; O=gpurun_out/r4f
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_parity.py -q -k "persistent_recurrence_matches or middle_first or recipe_shape" 2>&1 | tail -8 ) > $O/test_split.log 2>&1
one() { local label=$1; shift
  ( env "$@" 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); p=d.get('phase_ms_per_step',{})
        print('$label', round(d['ms_per_step'],2), 'fwd', round(p.get('recurrence_fwd',0),2), 'bwd', round(p.get('recurrence_bwd',0),2), 'in_gemm', round(p.get('input_gemm',0),2), flush=True)" ) >> $O/ab.log 2>&1; }
for round in 1 2 3; do
  one cfg2_f32rec      EESEN_FWD_SPLIT=0 python bench.py --main-only --steps 10 --warmup 3
  one cfg2_split       EESEN_FWD_SPLIT=1 python bench.py --main-only --steps 10 --warmup 3
  one cfg2_split_noov  EESEN_FWD_SPLIT=1 EESEN_OVERLAP=0 python bench.py --main-only --steps 10 --warmup 3
done
( EESEN_TRACE=1 EESEN_FWD_SPLIT=1 EESEN_OVERLAP=0 python bench.py --main-only --steps 2 --warmup 1 2>&1 | grep EESEN_TRACE ) > $O/trace.log 2>&1
cat $O/test_split.log $O/ab.log $O/trace.log
