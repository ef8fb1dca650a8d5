export EESEN_OVERLAP=0
for r in 1 2; do for x in 1 0; do
EESEN_FWD_XCHG=$x python bench.py --main-only --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']
print('xchg=$x', round(d['ms_per_step'],2), 'fwd', round(p['recurrence_fwd'],2), 'bwd', round(p['recurrence_bwd'],2))"
done; done
