#!/bin/bash
# Round 5, GPU call 2: CTC sweep with the measured wave defaults (+ the 2 x 2 arm at 256 positions), cfg2 at --num-sequence 64 under the
# schedule switches that could move it (side-stream occupancy cap, no overlap, the narrow forward tile as two workgroups per CU), its
# kernel timeline, and a parity check of the two-workgroups-per-CU forward grid.
mkdir -p gpurun_out/r5b; O=gpurun_out/r5b
export TMPDIR=/tmp
( timeout 300 python -m pytest -x -q tests/test_gpu_parity.py -k "ctc" 2>&1 | tail -5 ) > $O/tests_ctc.log 2>&1; cat $O/tests_ctc.log
( timeout 300 python scripts/ctc_waves_probe.py 2>$O/ctc_waves.err | tail -1 ) > $O/ctc_waves.json
python - <<'P'
import json
d = json.load(open("gpurun_out/r5b/ctc_waves.json"))
for k, v in d.items():
    print(k, "L'", v["Lprime"], {w: (round(x["us_per_lattice_step"], 3), x["identical_to_default"]) for w, x in v.items() if w.startswith("waves")})
P
one() { local label=$1; shift
  ( timeout 150 env "$@" python bench.py --main-only --S 64 --steps 6 --warmup 2 2>$O/s64_$label.err | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); p=d.get('phase_ms_per_step',{})
        print('$label', round(d['ms_per_step'],2), 'ms', round(d['value']), 'fps', {k: round(v,2) for k,v in p.items() if not k.startswith('ctc')}, flush=True)" ) >> $O/s64.log 2>&1
  grep -a -h "WARNING\|recover" $O/s64_$label.err | head -3 >> $O/s64.log; }
one default A=1
one side16 EESEN_SIDE_LDS_KB=16
one side0 EESEN_SIDE_LDS_KB=0
one nooverlap EESEN_OVERLAP=0
one narrow2 EESEN_FWD_NARROW2=1
one narrow2_m0 EESEN_FWD_NARROW2=1 EESEN_OCC_MARGIN=0
one narrow2_m0_side16 EESEN_FWD_NARROW2=1 EESEN_OCC_MARGIN=0 EESEN_SIDE_LDS_KB=16
one bwdq4off EESEN_BWD_Q4=0
cat $O/s64.log
( EESEN_FWD_NARROW2=1 EESEN_OCC_MARGIN=0 timeout 200 python - <<'P'
import os, numpy as np
from eesen_amd import synth
from eesen_amd.api import Net, Ctc
cfg = synth.config("cfg2"); cfg.update(S=64, T=120, layers=2)
layers = synth.make_model(**cfg); batch = synth.make_batch(**cfg)
def run():
    net = Net.from_layers(layers); net.SetTrainOptions(1e-3, 0.9); ctc = Ctc()
    net.SetSeqLengths(batch.lens); out = net.Propagate(batch.feats); d = ctc.EvalParallel(batch.lens, out, batch.labels); net.BackpropagateNoUpdate(d)
    g = net.GetGrads(); info = net.RecurrenceInfo(); return out.numpy(), g, info, net.recoveries
a = run()
os.environ["EESEN_FWD_SPLIT"] = "0"     # the fp32-input narrow tile (bit-identical to the per-step path) as the arbiter
b = run()
rel = lambda x, y: float(np.max(np.abs(x - y)) / np.max(np.abs(y)))
print("two-workgroups-per-CU forward grid at S=64: info", a[2], "recoveries", a[3], "out vs fp32 tile", rel(a[0], b[0]), "grads", rel(a[1], b[1]), "arbiter info", b[2])
P
) > $O/narrow2_parity.log 2>&1; tail -3 $O/narrow2_parity.log
R=$GRAFT_REPO_ROOT
cd /tmp && ( timeout 200 rocprofv3 --kernel-trace -d $R/$O/prof -o s64 -- python $R/bench.py --steps 3 --warmup 1 --main-only --S 64 > $R/$O/prof.log 2>&1 )
cd $R
DB=$(find $O/prof -name "*_results.db" | head -1)
if [ -n "$DB" ]; then
  python scripts/rocpd_summary.py $DB > $O/r05_s64_kernel_stats.md
  python scripts/timeline.py $DB > $O/r05_s64_step_timeline.txt 2>/dev/null
  head -14 $O/r05_s64_kernel_stats.md; cat $O/r05_s64_step_timeline.txt | head -70
  rm -rf $O/prof
fi
