#!/bin/bash
# Round 6, call 1: the refactored selection (RecPlan) under the tests that exercise it, the new multi-rank tests with the RCCL-shaped
# stand-in, the self-describing bench line, a short headline bench.
mkdir -p gpurun_out/r6a; O=gpurun_out/r6a
export TMPDIR=/tmp EESEN_PARITY_OUT=$PWD/$O
( timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_parallel.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -40 ) > $O/test_multi.log 2>&1; cat $O/test_multi.log
( timeout 300 python bench.py --main-only --steps 10 --warmup 3 2>$O/bench_err.log | cut -c1-3000 ) > $O/bench_main.log; cat $O/bench_main.log | head -c 2500; tail -5 $O/bench_err.log
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -8 ) > $O/test_parity.log 2>&1; cat $O/test_parity.log
