#!/bin/bash
# Round 4, last GPU call: EESEN_BWD_EARLY (the backward recurrence's cell operands requested at the top of the step) -- bit identity,
# same-box A/B at cfg2 and cfg4, in-kernel timelines of both arms, first-poll delay sweep, then the parity / recovery tests under it.
mkdir -p gpurun_out/r4g; O=gpurun_out/r4g
export TMPDIR=/tmp
( timeout 240 python -m pytest tests/test_gpu_parity.py -q -x -k "early_cell or backward_tiles" 2>&1 | tail -6 ) > $O/test_early.log 2>&1
cat $O/test_early.log
one() { local label=$1; shift
  ( timeout 120 env "$@" 2>$O/last_stderr.log | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); p=d.get('phase_ms_per_step',{}); r=d.get('roofline',{})
        print('$label', round(d['ms_per_step'],2), 'fwd', round(p.get('recurrence_fwd',0),2), 'bwd', round(p.get('recurrence_bwd',0),2), 'grad_gemm', round(p.get('grad_gemm',0),2),
              'dom', r.get('kernel'), round(r.get('avg_launch_us',0),1), 'alone', round((r.get('alone') or {}).get('avg_launch_us',0),1), flush=True)" ) >> $O/ab.log 2>&1; }
for round in 1 2; do
  one cfg2_late   EESEN_BWD_EARLY=0 python bench.py --main-only --steps 10 --warmup 3
  one cfg2_early  EESEN_BWD_EARLY=1 python bench.py --main-only --steps 10 --warmup 3
done
cat $O/ab.log
# in-kernel timelines, nothing overlapped (EESEN_OVERLAP=0): wait | fetch+MFMA+reduce | cell | drain | publish->next
for m in 0 1; do
  ( EESEN_BWD_EARLY=$m EESEN_TRACE=1 EESEN_OVERLAP=0 EESEN_PRINT_FLIGHT=1 timeout 100 python bench.py --main-only --steps 2 --warmup 1 2>&1 | grep -a "EESEN_TRACE\|increment flight" | sed "s/^/early=$m /" ) >> $O/trace.log 2>&1
done
cat $O/trace.log
# first-poll delay of the backward wait under EARLY (the forward delay stays what the box derives)
F=$(grep -a -m1 "increment flight" $O/trace.log | sed -E 's/.*forward ([0-9]+), backward ([0-9]+) ns.*/\1/')
if [ -n "$F" ]; then
  for b in 100 200 450 600; do
    one cfg2_early_bwd_delay_$b EESEN_BWD_EARLY=1 EESEN_POLL_NS="$F,$b" python bench.py --main-only --steps 10 --warmup 3
  done
fi
one cfg4_late   EESEN_BWD_EARLY=0 python bench.py --config cfg4 --main-only --steps 5 --warmup 2
one cfg4_early  EESEN_BWD_EARLY=1 python bench.py --config cfg4 --main-only --steps 5 --warmup 2
one cfg4_late   EESEN_BWD_EARLY=0 python bench.py --config cfg4 --main-only --steps 5 --warmup 2
one cfg4_early  EESEN_BWD_EARLY=1 python bench.py --config cfg4 --main-only --steps 5 --warmup 2
for m in 0 1; do
  ( EESEN_BWD_EARLY=$m EESEN_TRACE=1 timeout 100 python bench.py --config cfg4 --main-only --steps 2 --warmup 1 2>&1 | grep -a "EESEN_TRACE" | sed "s/^/cfg4 early=$m /" ) >> $O/trace.log 2>&1
done
tail -n +5 $O/ab.log; tail -4 $O/trace.log
# the parity, odd-shape, recipe-shape, give-up and recovery tests with the new arm selected
( EESEN_BWD_EARLY=1 timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_comm.py -q -x 2>&1 | tail -6 ) > $O/test_parity_early.log 2>&1
cat $O/test_parity_early.log
( EESEN_BWD_EARLY=1 timeout 400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_reference_fullsize.py -q -x -k "cfg2 or cfg4" 2>&1 | tail -6 ) > $O/test_full_early.log 2>&1
cat $O/test_full_early.log
