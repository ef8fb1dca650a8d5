run() { # name
  L=""; [ "$1" != base ] && L="eesen_amd/lib/variants/libeesen_hip_$1.so"
  for mode in full alone; do
    E=""; [ $mode = alone ] && E="EESEN_OVERLAP=0 EESEN_GATE_FWD=0"
    env EESEN_HIP_LIBRARY=$L $E python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('$1 $mode', round(d['ms_per_step'],2), round(p['recurrence_fwd'],2), round(p['recurrence_bwd'],2))"
  done
}
for v in "$@"; do run $v; done
