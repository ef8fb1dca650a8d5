#!/bin/bash
# In-kernel timelines of the default kernels UNDER LOAD (side-stream GEMMs beside the backward recurrence) and alone, same box.
mkdir -p gpurun_out/r4i; O=gpurun_out/r4i
export TMPDIR=/tmp
for ov in 1 0; do
  ( EESEN_TRACE=1 EESEN_OVERLAP=$ov timeout 100 python bench.py --main-only --steps 3 --warmup 1 2>&1 | grep -a "EESEN_TRACE" | sed "s/^/overlap=$ov /" ) >> $O/trace.log 2>&1
done
( EESEN_TRACE=1 EESEN_FWD_MID=0 timeout 100 python bench.py --main-only --steps 3 --warmup 1 2>&1 | grep -a "EESEN_TRACE" | sed "s/^/overlap=1 fwd_mid=0 /" ) >> $O/trace.log 2>&1
cat $O/trace.log
