#!/bin/bash
# Round 5, GPU call 9: the deferred bucket schedule (EESEN_COMM_DEFER=1) through the communicator tests.
mkdir -p gpurun_out/r5i; O=gpurun_out/r5i
export TMPDIR=/tmp EESEN_PARITY_OUT=$PWD/$O
( timeout 900 python -m pytest -x -q tests/test_gpu_comm.py tests/test_gpu_multirank.py tests/test_gpu_parallel.py 2>&1 | tail -6 ) > $O/tests.log 2>&1; cat $O/tests.log
