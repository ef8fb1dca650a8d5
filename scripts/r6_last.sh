#!/bin/bash
# Round 6, last call: the whole GPU suite, smoke() and the driver's default bench command on the committed tree.
mkdir -p gpurun_out/r6last; O=gpurun_out/r6last
export TMPDIR=/tmp EESEN_PARITY_OUT=$PWD/$O
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/test_gpu.log 2>&1; cat $O/test_gpu.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/smoke.log; cat $O/smoke.log
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; python -c "
import json
d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][0])
print('default bench:', d['metric'], round(d['value']), d['unit'], 'ms/step', round(d['ms_per_step'],2), 'steps', d['steps'], 'warmup', d['warmup'], 'roofline frac', round(d['roofline']['frac'],3), 'stale', d['roofline']['traffic_source']['stale'], 'cpu', round(d['cpu_baseline']['value']))
print(d['legs_ms'])"
