#!/bin/bash
# the driver's command once more, with profiles/pmc_traffic.json of THIS tree in place (traffic_source.stale must read false), and the
# round-5 GEMM arithmetic as a separate process for comparison with the in-process leg config.bf16_split_gemm
mkdir -p gpurun_out/r6y; O=gpurun_out/r6y
export TMPDIR=/tmp
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_r06.json 2> $O/bench.err; tail -c 600 $O/bench_r06.json
( EESEN_GEMM_MODE=split timeout 200 python bench.py --main-only --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('standalone EESEN_GEMM_MODE=split (fp16-plane recurrences)', round(d['ms_per_step'],3), 'ms', flush=True)" ) > $O/split_standalone.log; cat $O/split_standalone.log
( EESEN_GEMM_MODE=split EESEN_FWD_F16=0 timeout 200 python bench.py --main-only --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('standalone EESEN_GEMM_MODE=split EESEN_FWD_F16=0 (round 5 arithmetic)', round(d['ms_per_step'],3), 'ms', flush=True)" ) >> $O/split_standalone.log; cat $O/split_standalone.log
