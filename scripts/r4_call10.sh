#!/bin/bash
# cell counts that are not multiples of 4 padded inside the library: the new tests, then everything that goes through parameter I/O
mkdir -p gpurun_out/r4k
( timeout 100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -x 2>&1 | tail -12 ) > gpurun_out/r4k/parity.log 2>&1; cat gpurun_out/r4k/parity.log
( timeout 80 python -m pytest tests/test_gpu_cli.py tests/test_gpu_dropout.py -q -x 2>&1 | tail -6 ) > gpurun_out/r4k/cli.log 2>&1; cat gpurun_out/r4k/cli.log
