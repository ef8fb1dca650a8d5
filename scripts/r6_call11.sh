#!/bin/bash
# Round 6, call 11: where the fp16-plane K-split backward step goes (in-kernel timeline, probes); GEMM accuracy tests after the arena fix.
mkdir -p gpurun_out/r6k; O=gpurun_out/r6k
export TMPDIR=/tmp EESEN_PARITY_OUT=$PWD/$O
( timeout 600 python -m pytest tests/test_gpu_gemm.py -q 2>&1 | tail -8 ) > $O/test_gemm.log 2>&1; cat $O/test_gemm.log
run() {  # label, env...
  local label=$1; shift
  ( env "$@" timeout 300 python bench.py --config cfg4 --main-only --steps 5 --warmup 2 2>$O/$label.err | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); p=d['phase_ms_per_step']
        print('$label', 'step', round(d['ms_per_step'],2), 'ms; recurrence_bwd', round(p['recurrence_bwd'],2), 'fwd', round(p['recurrence_fwd'],2), 'grad_gemm', round(p['grad_gemm'],2), flush=True)" ) >> $O/ledger.log 2>&1
  grep EESEN_TRACE $O/$label.err | tail -3 >> $O/ledger.log
}
run bwd_f32 EESEN_BWD_F16=0 EESEN_TRACE=1
run bwd_f16 EESEN_BWD_F16=1 EESEN_TRACE=1
run bwd_f16_again EESEN_BWD_F16=1
for n in 4 7; do run probe$n EESEN_HIP_LIBRARY=$PWD/eesen_amd/lib/variants/libeesen_hip_khprobe$n.so EESEN_TRACE=1; done
cat $O/ledger.log
