#!/bin/bash
# Round 5, GPU call 10: the two full-size arbiter cases added last -- cfg2's net at --num-sequence 64 and the recipes' 320-cell width at
# full length -- against the fixtures made from the reference (oracle/fullsize.py).
mkdir -p gpurun_out/r5j; O=gpurun_out/r5j
export TMPDIR=/tmp EESEN_PARITY_OUT=$PWD/$O
( timeout 600 python -m pytest -x -q tests/test_gpu_reference_fullsize.py -k "num_sequence_64 or recipe_width" 2>&1 | tail -12 ) > $O/tests.log 2>&1; cat $O/tests.log
