#!/bin/bash
mkdir -p gpurun_out/r6h; O=gpurun_out/r6h
export TMPDIR=/tmp EESEN_PARITY_OUT=$PWD/$O
( timeout 300 python -m pytest tests/test_gpu_gemm.py -q -k "bounds_pass or keep_what" 2>&1 | tail -15 ) > $O/test_bounds.log 2>&1; cat $O/test_bounds.log
( timeout 300 python scripts/debug_half.py 2>&1 | tail -60 ) > $O/debug.log; cat $O/debug.log
