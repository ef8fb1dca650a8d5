#!/bin/bash
mkdir -p gpurun_out/r6u; O=gpurun_out/r6u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && ( timeout 200 rocprofv3 --kernel-trace -d $R/$O/prof -o w -- python $R/bench.py --steps 3 --warmup 2 --main-only > $R/$O/prof.log 2>&1 )
cd $R
DB=$(find $O/prof -name "*_results.db" | head -1)
python scripts/timeline_window.py $DB > $O/window.txt; rm -rf $O/prof
wc -l $O/window.txt; grep -n "lstm_fwd_persistent\|lstm_bwd_persistent" $O/window.txt | head
