#!/bin/bash
# Round 6, call 14: GEMM on operands split ahead of the call -- bit-identity with the in-kernel split, then the rates
mkdir -p gpurun_out/r6p; O=gpurun_out/r6p
export TMPDIR=/tmp
( timeout 600 python scripts/gemm_pre_probe.py check 2>&1 | tail -60 ) > $O/check.log; grep -c bit-identical $O/check.log; grep DIFFERS $O/check.log | head
( timeout 900 python scripts/gemm_pre_probe.py bench 2>&1 | tail -40 ) > $O/bench.log; cat $O/bench.log
