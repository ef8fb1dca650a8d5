#!/bin/bash
# Round 4, closing GPU call: the whole GPU suite on the final tree, the driver's bench line, the forward first-poll delay either side of
# the derived one, and a kernel trace of the same bench command.
mkdir -p gpurun_out/r4h; O=gpurun_out/r4h
export TMPDIR=/tmp
( timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/test_gpu.log 2>&1
cat $O/test_gpu.log
( timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_stderr.log | tail -1 ) > $O/bench_line.json
python - <<'P'
import json
try:
    d = json.load(open("gpurun_out/r4h/bench_line.json"))
    print("bench", round(d["ms_per_step"], 2), "ms", round(d["value"]), d["unit"], "frac", round(d["roofline"]["frac"], 4),
          {k: round(v.get("ms_per_step", v.get("ms_per_minibatch", 0)), 1) for k, v in d["config"].get("secondary", {}).items() if isinstance(v, dict)})
except Exception as e:
    print("bench line unreadable:", e)
P
one() { local label=$1; shift
  ( timeout 120 env "$@" 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); p=d.get('phase_ms_per_step',{})
        print('$label', round(d['ms_per_step'],2), 'fwd', round(p.get('recurrence_fwd',0),2), 'bwd', round(p.get('recurrence_bwd',0),2), flush=True)" ) >> $O/ab.log 2>&1; }
F=$(EESEN_PRINT_FLIGHT=1 timeout 100 python bench.py --main-only --steps 1 --warmup 1 2>&1 | grep -a -m1 "increment flight")
echo "$F" >> $O/ab.log
FW=$(echo "$F" | sed -E 's/.*forward ([0-9]+), backward ([0-9]+) ns.*/\1/'); BW=$(echo "$F" | sed -E 's/.*forward ([0-9]+), backward ([0-9]+) ns.*/\2/')
if [ -n "$FW" ] && [ -n "$BW" ]; then
  for f in $((FW*2/3)) $FW $((FW*4/3)); do one cfg2_fwd_delay_$f EESEN_POLL_NS="$f,$BW" python bench.py --main-only --steps 10 --warmup 3; done
fi
cat $O/ab.log
R=$GRAFT_REPO_ROOT
cd /tmp && ( timeout 200 rocprofv3 --kernel-trace -d $R/$O/prof -o r04b -- python $R/bench.py --steps 3 --warmup 1 --main-only > $R/$O/prof.log 2>&1 )
cd $R
DB=$(find $O/prof -name "*_results.db" | head -1)
if [ -n "$DB" ]; then
  python scripts/rocpd_summary.py $DB > $O/r04b_kernel_stats.md
  python scripts/timeline.py $DB > $O/r04b_step_timeline.txt 2>/dev/null
  head -16 $O/r04b_kernel_stats.md
  rm -rf $O/prof
fi
