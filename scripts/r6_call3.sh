#!/bin/bash
# Round 6, call 3: the K-split backward kernel's ledger for "W_m^T as three bf16 planes" (VERDICT r5 item 4): the in-kernel timeline of
# the product kernel at cfg4, and the step time of three costing probes (MFMA chain at the bf16 form's length; operand fetch grown
# by half; both).  Results of the probes are garbage by construction -- only the backward recurrence's time is read.
mkdir -p gpurun_out/r6c; O=gpurun_out/r6c
export TMPDIR=/tmp
run() {  # label, env...
  local label=$1; shift
  ( env "$@" timeout 300 python bench.py --config cfg4 --forward-precision bf16 --main-only --steps 5 --warmup 2 2>$O/$label.err | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); p=d['phase_ms_per_step']
        print('$label', 'step', round(d['ms_per_step'],2), 'ms; recurrence_bwd', round(p['recurrence_bwd'],2), 'fwd', round(p['recurrence_fwd'],2), 'grad_gemm', round(p['grad_gemm'],2), 'input_gemm', round(p['input_gemm'],2), flush=True)" ) >> $O/ledger.log 2>&1
  grep EESEN_TRACE $O/$label.err | tail -4 >> $O/ledger.log
}
run product EESEN_TRACE=1
run product_again
for n in 1 2 3; do run probe$n EESEN_HIP_LIBRARY=$PWD/eesen_amd/lib/variants/libeesen_hip_ksprobe$n.so EESEN_TRACE=1; done
run product_f32 EESEN_GEMM_MODE=split
cat $O/ledger.log
