#!/bin/bash
mkdir -p gpurun_out/r6q; O=gpurun_out/r6q
export TMPDIR=/tmp
for m in a b c d e; do ( echo "== mode $m"; timeout 300 python scripts/placement_probe.py $m 2>&1 | tail -8 ) >> $O/placement.log; done; cat $O/placement.log
