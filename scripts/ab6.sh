# schedule knobs re-checked after the recurrence changes (same box, cfg2, 10 steps each)
run() { env "$@" python bench.py --main-only --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']
print('$*', round(d['ms_per_step'],2), 'fwd', round(p['recurrence_fwd'],2), 'bwd', round(p['recurrence_bwd'],2))"; }
for r in 1 2; do
run X=0
run EESEN_SIDE_LDS_KB=32
run EESEN_SIDE_LDS_KB=64
run EESEN_GATE_FWD=1
run EESEN_GATE_FWD=1 EESEN_SIDE_LDS_KB=32
run EESEN_BWD_SEQ_TILE=16
done
