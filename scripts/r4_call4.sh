#!/bin/bash
# round 4, GPU call 4: the 4 x 32 forward tile (parity, A/B) and the two-plane W_m variant of the bf16 recurrence
mkdir -p gpurun_out/r4d; O=gpurun_out/r4d
export TMPDIR=/tmp
( EESEN_FWD_Q4=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -x -k "train_step_parity or golden or odd_shapes or middle_first or recipe_shape or projection or momentum" 2>&1 | tail -15 ) > $O/test_q4f.log 2>&1
one() { local label=$1; shift
  ( env "$@" 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); p=d.get('phase_ms_per_step',{})
        print('$label', round(d['ms_per_step'],2), 'fwd', round(p.get('recurrence_fwd',0),2), 'bwd', round(p.get('recurrence_bwd',0),2), 'in_gemm', round(p.get('input_gemm',0),2), flush=True)" ) >> $O/ab.log 2>&1; }
for round in 1 2; do
  one cfg2_base      EESEN_FWD_Q4=0 python bench.py --main-only --steps 10 --warmup 3
  one cfg2_q4f       EESEN_FWD_Q4=1 python bench.py --main-only --steps 10 --warmup 3
  one cfg2_base_noov EESEN_FWD_Q4=0 EESEN_OVERLAP=0 python bench.py --main-only --steps 10 --warmup 3
  one cfg2_q4f_noov  EESEN_FWD_Q4=1 EESEN_OVERLAP=0 python bench.py --main-only --steps 10 --warmup 3
  one cfg4_bf16_w1   python bench.py --config cfg4 --main-only --steps 5 --warmup 2 --forward-precision bf16
  one cfg4_bf16_w2   EESEN_BF16_REC_WPLANES=2 python bench.py --config cfg4 --main-only --steps 5 --warmup 2 --forward-precision bf16
done
( EESEN_TRACE=1 EESEN_FWD_Q4=1 EESEN_OVERLAP=0 python bench.py --main-only --steps 2 --warmup 1 2>&1 | grep EESEN_TRACE ) > $O/trace_q4f.log 2>&1
( EESEN_TRACE=1 EESEN_FWD_Q4=0 EESEN_OVERLAP=0 python bench.py --main-only --steps 2 --warmup 1 2>&1 | grep EESEN_TRACE ) >> $O/trace_q4f.log 2>&1
( EESEN_BF16_REC_WPLANES=2 EESEN_PARITY_OUT=$O timeout 1200 python -m pytest tests/test_gpu_reference_fullsize.py -q -k "bf16_forward_variant" 2>&1 | tail -6 ) > $O/test_w2.log 2>&1
cat $O/test_q4f.log $O/ab.log $O/trace_q4f.log $O/test_w2.log
python - <<'PY'
import json
for r in json.load(open('gpurun_out/r4d/parity_fullsize.json')):
    print(r['case'], 'grads_worst', r.get('grads_worst'), 'lnp', r.get('ln_p_rel_err_per_sequence'), 'net_out', r.get('net_out_valid'))
PY
