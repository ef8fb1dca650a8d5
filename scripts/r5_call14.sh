#!/bin/bash
# Round 5, GPU call 14: net-output-extract over the synthetic table, one utterance at a time against 32 together.
mkdir -p gpurun_out/r5n; O=gpurun_out/r5n
export TMPDIR=/tmp
( timeout 600 python scripts/trainer_throughput.py --num-sequence 32 --frame-limit 100000 --extract 2>$O/err.log | tail -1 ) > $O/extract.json; cat $O/extract.json; tail -2 $O/err.log
