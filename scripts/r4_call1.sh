#!/bin/bash
# round 4, GPU call 1: the bf16 forward recurrence (parity vs its numpy arbiter, cfg4 A/B) and two-chains-per-workgroup at cfg2
mkdir -p gpurun_out/r4a
O=gpurun_out/r4a
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_bf16_forward.py -x -q 2>&1 | tail -15 ) > $O/test_bf16.log 2>&1
one() { # label, env..., -- bench args
  local label=$1; shift
  ( env "$@" 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line); p=d.get('phase_ms_per_step',{})
    print('$label', round(d['ms_per_step'],2), 'fwd', round(p.get('recurrence_fwd',0),2), 'bwd', round(p.get('recurrence_bwd',0),2), 'gemm_in', round(p.get('input_gemms',0),2), flush=True)
" ) >> $O/ab.log 2>&1
}
for round in 1 2; do
  one cfg2_base       EESEN_FWD_MUX2=0 python bench.py --main-only --steps 10 --warmup 3
  one cfg2_mux2_4u    EESEN_FWD_MUX2=1 python bench.py --main-only --steps 10 --warmup 3
  one cfg2_mux2_8u    EESEN_FWD_MUX2=2 python bench.py --main-only --steps 10 --warmup 3
  one cfg2_base_noov  EESEN_FWD_MUX2=0 EESEN_OVERLAP=0 python bench.py --main-only --steps 10 --warmup 3
  one cfg2_mux2_4u_noov EESEN_FWD_MUX2=1 EESEN_OVERLAP=0 python bench.py --main-only --steps 10 --warmup 3
  one cfg2_mux2_8u_noov EESEN_FWD_MUX2=2 EESEN_OVERLAP=0 python bench.py --main-only --steps 10 --warmup 3
  one cfg4_f32        python bench.py --config cfg4 --main-only --steps 5 --warmup 2
  one cfg4_bf16       python bench.py --config cfg4 --main-only --steps 5 --warmup 2 --forward-precision bf16
  one cfg4_bf16gemm   python bench.py --config cfg4 --main-only --steps 5 --warmup 2 --forward-precision bf16-gemm
done
( EESEN_FWD_MUX2=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "test_persistent_recurrence_matches_step_kernels or test_middle_first" 2>&1 | tail -5 ) > $O/test_mux2_1.log 2>&1
( EESEN_FWD_MUX2=2 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "test_persistent_recurrence_matches_step_kernels or test_middle_first" 2>&1 | tail -5 ) > $O/test_mux2_2.log 2>&1
( EESEN_TRACE=1 python bench.py --config cfg4 --main-only --steps 2 --warmup 1 --forward-precision bf16 2>&1 | grep EESEN_TRACE ) > $O/trace_cfg4_bf16.log 2>&1
cat $O/test_bf16.log $O/ab.log $O/test_mux2_1.log $O/test_mux2_2.log $O/trace_cfg4_bf16.log
