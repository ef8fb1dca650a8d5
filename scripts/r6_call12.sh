#!/bin/bash
# Round 6, call 12: the fp16-plane K-split tile with K split EIGHT ways (EESEN_BWD_K8): parity, then the cfg4 / cfg5 steps and the in-kernel timeline.
mkdir -p gpurun_out/r6l; O=gpurun_out/r6l
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "wide_backward_tile" 2>&1 | tail -8 ) > $O/test.log 2>&1; cat $O/test.log
run() {
  local label=$1; shift
  ( env "$@" timeout 400 python bench.py --main-only --steps 5 --warmup 2 $CFG 2>$O/$label.err | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); p=d['phase_ms_per_step']
        print('$label', 'step', round(d['ms_per_step'],2), 'ms; recurrence_bwd', round(p['recurrence_bwd'],2), 'fwd', round(p['recurrence_fwd'],2), 'grad_gemm', round(p['grad_gemm'],2), d['config']['kernels']['backward'], flush=True)" ) >> $O/ledger.log 2>&1
  grep "EESEN_TRACE bwd" $O/$label.err | tail -2 >> $O/ledger.log
}
CFG="--config cfg4"
run cfg4_k4 EESEN_BWD_K8=0 EESEN_TRACE=1
run cfg4_k8 EESEN_BWD_K8=1 EESEN_TRACE=1
run cfg4_k4_again EESEN_BWD_K8=0
run cfg4_k8_again EESEN_BWD_K8=1
CFG="--config cfg5 --steps 3 --warmup 1"
run cfg5_k4 EESEN_BWD_K8=0
run cfg5_k8 EESEN_BWD_K8=1
cat $O/ledger.log
