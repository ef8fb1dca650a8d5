#!/bin/bash
# Round 5, GPU call 11: the trainer binaries end to end on a synthetic archive (native C++ and Python), recipe shape.
mkdir -p gpurun_out/r5k; O=gpurun_out/r5k
export TMPDIR=/tmp
for S in 32 10; do
  ( timeout 400 python scripts/trainer_throughput.py --num-sequence $S --frame-limit $([ $S = 10 ] && echo 25000 || echo 100000) 2>$O/err_$S.log | tail -1 ) > $O/trainer_S$S.json; cat $O/trainer_S$S.json; tail -2 $O/err_$S.log
done
