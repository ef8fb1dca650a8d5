#!/bin/bash
# is the bimodal step time of a fresh process tied to the measured increment flight (first-poll delays)?
mkdir -p gpurun_out/r6q; O=gpurun_out/r6q
export TMPDIR=/tmp
run() {  # tag, env...
  local tag=$1; shift
  for i in 1 2 3 4 5 6; do
    ( env "$@" EESEN_PRINT_FLIGHT=1 timeout 120 python bench.py --main-only --steps 10 --warmup 3 2>$O/err.txt | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('$tag run $i', round(d['ms_per_step'],2), 'ms', end=' ')" ; grep -h "increment flight" $O/err.txt | head -1 ) >> $O/flight.log
  done
}
run split EESEN_GEMM_MODE=split
run split_fixed EESEN_GEMM_MODE=split EESEN_POLL_NS=600,420
run half
cat $O/flight.log
