#!/bin/bash
# Round 5, GPU call 12: device buffers that grow by half instead of exactly (lists sorted by length: T rises every minibatch) -- the
# trainer binaries end to end, and the tests whose minibatch shapes change from step to step.
mkdir -p gpurun_out/r5l; O=gpurun_out/r5l
export TMPDIR=/tmp
for S in 32 10; do
  ( timeout 400 python scripts/trainer_throughput.py --num-sequence $S --frame-limit $([ $S = 10 ] && echo 25000 || echo 100000) 2>$O/err_$S.log | tail -1 ) > $O/trainer_S$S.json; cat $O/trainer_S$S.json; tail -2 $O/err_$S.log
done
( timeout 400 python scripts/trainer_throughput.py --num-sequence 32 --frame-limit 100000 --utts 2048 2>$O/err_2048.log | tail -1 ) > $O/trainer_S32_2048.json; cat $O/trainer_S32_2048.json
( timeout 600 python -m pytest -x -q tests/test_gpu_cli.py tests/test_gpu_feeder.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_frontend.py 2>&1 | tail -5 ) > $O/tests.log 2>&1; cat $O/tests.log
