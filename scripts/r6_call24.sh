#!/bin/bash
# Round 6: the trainer BINARIES end to end on a length-sorted Kaldi archive (round 5's record: 808 k / 1.02 M padded frames/s at --num-sequence 32)
mkdir -p gpurun_out/r6t; O=gpurun_out/r6t
export TMPDIR=/tmp
for S in 10 32 64; do
  ( timeout 400 python scripts/trainer_throughput.py --num-sequence $S --frame-limit $([ $S = 10 ] && echo 25000 || echo 100000) 2>$O/err_$S.log | tail -1 ) > $O/trainer_S$S.json; cut -c1-600 $O/trainer_S$S.json
done
( timeout 600 python scripts/trainer_throughput.py --num-sequence 32 --frame-limit 100000 --utts 2048 2>$O/err_2048.log | tail -1 ) > $O/trainer_S32_2048.json; cut -c1-600 $O/trainer_S32_2048.json
( timeout 600 python scripts/trainer_throughput.py --num-sequence 64 --frame-limit 200000 --utts 2048 2>$O/err_2048b.log | tail -1 ) > $O/trainer_S64_2048.json; cut -c1-600 $O/trainer_S64_2048.json
