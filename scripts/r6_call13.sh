#!/bin/bash
# Round 6, call 13: the recurrence kernels' cross-wave LDS reductions with all reads in flight (forward plane kernel, 4 x 32 backward tile,
# fp16-plane K-split tile): the cfg2 / cfg4 steps and the in-kernel timelines; results must be bit-identical to the build before (same order).
mkdir -p gpurun_out/r6m; O=gpurun_out/r6m
export TMPDIR=/tmp
run() {
  local label=$1; shift
  ( env "$@" timeout 400 python bench.py --main-only --steps 20 --warmup 5 $CFG 2>$O/$label.err | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); p=d['phase_ms_per_step']
        print('$label', 'step', round(d['ms_per_step'],3), 'ms; recurrence_fwd', round(p['recurrence_fwd'],2), 'bwd', round(p['recurrence_bwd'],2), flush=True)" ) >> $O/ledger.log 2>&1
  grep "EESEN_TRACE" $O/$label.err | tail -2 >> $O/ledger.log
}
CFG=""
run cfg2_a; run cfg2_b; run cfg2_trace EESEN_TRACE=1
CFG="--config cfg4 --steps 10"
run cfg4_a; run cfg4_trace EESEN_TRACE=1
cat $O/ledger.log
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "persistent_recurrence_matches or wide_backward or backward_tiles or train_step" 2>&1 | tail -4 ) > $O/test.log 2>&1; cat $O/test.log
