#!/bin/bash
mkdir -p gpurun_out/r6p; O=gpurun_out/r6p
export TMPDIR=/tmp
( GEMM_BENCH_ONLY=cfg2 timeout 900 python scripts/gemm_pre_probe.py bench 2>&1 | tail -40 ) > $O/bench2.log; cat $O/bench2.log
( GEMM_BENCH_ONLY=cfg4 timeout 900 python scripts/gemm_pre_probe.py bench 2>&1 | tail -40 ) > $O/bench4.log; cat $O/bench4.log
