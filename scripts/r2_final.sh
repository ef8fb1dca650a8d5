#!/bin/bash
# One consolidated GPU call: the whole -m gpu suite, smoke(), the default bench line.
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/pytest_gpu_full.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu_full.log
tail -5 gpurun_out/pytest_gpu_full.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench exit $?"
cat gpurun_out/bench_final.json
