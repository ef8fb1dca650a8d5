#!/bin/bash
# Round 6, call 6: two fp16 planes as the default GEMM arithmetic AND in the forward recurrence (narrow and wide tiles): the whole GPU
# suite, then the steps with EESEN_FWD_F16=0 / 1 on the same box.
mkdir -p gpurun_out/r6f; O=gpurun_out/r6f
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
for r in 1 2; do
  EESEN_FWD_F16=0 run cfg2_bf16planes
  EESEN_FWD_F16=1 run cfg2_f16planes
done
EESEN_FWD_F16=0 run cfg4_fwd_f32mfma --config cfg4
EESEN_FWD_F16=1 run cfg4_fwd_f16planes --config cfg4
EESEN_FWD_F16=0 run cfg5_fwd_f32mfma --config cfg5 --steps 3 --warmup 1
EESEN_FWD_F16=1 run cfg5_fwd_f16planes --config cfg5 --steps 3 --warmup 1
EESEN_FWD_F16=0 run cfg2S64_bf16planes --S 64
EESEN_FWD_F16=1 run cfg2S64_f16planes --S 64
cat $O/ab.log
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/test_gpu.log 2>&1; cat $O/test_gpu.log
