#!/bin/bash
# Round 5, GPU call 8: boundary row blocks zeroed only when the shape changed -- same-box A/B against the always-zero variant
# (scripts/build_variant.py zeroalways "-DEESEN_ZERO_ALWAYS=1" net.cpp), and the tests whose minibatch shapes change from step to step.
mkdir -p gpurun_out/r5h; O=gpurun_out/r5h
export TMPDIR=/tmp
bash scripts/ab_variants.sh "--steps 20 --warmup 5" zeroalways > $O/ab.log 2>&1; cat $O/ab.log
( timeout 600 python -m pytest -x -q tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_feeder.py tests/test_gpu_dropout.py 2>&1 | tail -6 ) > $O/tests.log 2>&1; cat $O/tests.log
