#!/bin/bash
# Round 6, call 2: the --check full_cfg3 leg with eight ranks through the stand-in; the 8-rank bench test with the share worked out by
# the communicator; final-tree counters for the wide configurations (cfg4 fp32, cfg4 bf16-forward, cfg5).
mkdir -p gpurun_out/r6b; O=gpurun_out/r6b
export TMPDIR=/tmp EESEN_PARITY_OUT=$PWD/$O
( timeout 1500 python -m pytest tests/test_gpu_parallel.py -x -q 2>&1 | tail -30 ) > $O/test_parallel.log 2>&1; cat $O/test_parallel.log
bash scripts/collect_profiles_wide.sh r06 > $O/collect_wide.log 2>&1; tail -90 $O/collect_wide.log
