#!/bin/bash
# round 4, GPU call 3: tests touched this round + A/B of the command-processor wait arm
mkdir -p gpurun_out/r4c; O=gpurun_out/r4c
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_comm.py tests/test_gpu_parallel.py tests/test_gpu_bf16_forward.py tests/test_gpu_cli.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -25 ) > $O/tests_a.log 2>&1
( EESEN_PARITY_OUT=$O timeout 1500 python -m pytest tests/test_gpu_reference_fullsize.py -q -k "cfg2 or cfg4" 2>&1 | tail -25 ) > $O/tests_full.log 2>&1
for round in 1 2; do
  for mid in 1 2 0; do
    ( EESEN_FWD_MID=$mid python bench.py --main-only --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); p=d['phase_ms_per_step']
        print('fwd_mid=$mid', round(d['ms_per_step'],2), 'fwd', round(p['recurrence_fwd'],2), 'bwd', round(p['recurrence_bwd'],2))" ) >> $O/ab.log 2>&1
  done
done
cat $O/tests_a.log $O/tests_full.log $O/ab.log
