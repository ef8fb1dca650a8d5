# A/B harness: bash scripts/ab.sh [-t] base variantA base:ENV=1,ENV2=0 ...   (variants built by scripts/build_variant.py)
# prints ms/step, forward and backward recurrence phase ms for the full step and for the no-overlap ("alone") schedule
TR=0; [ "$1" = "-t" ] && { TR=1; shift; }
run() {
  name=${1%%:*}; envs=""; [ "$1" != "$name" ] && envs=$(echo "${1#*:}" | tr ',' ' ')
  L=""; [ "$name" != base ] && L="eesen_amd/lib/variants/libeesen_hip_$name.so"
  for mode in full alone; do
    E=""; [ $mode = alone ] && E="EESEN_OVERLAP=0 EESEN_GATE_FWD=0"
    env EESEN_HIP_LIBRARY=$L EESEN_TRACE=$TR $E $envs python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/tmp/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('$1 $mode', round(d['ms_per_step'],2), round(p['recurrence_fwd'],2), round(p['recurrence_bwd'],2))"
    [ $TR = 1 ] && grep "EESEN_TRACE [fb]wd:" /tmp/ab.err | sed 's/ticks.*//'
  done
}
for v in "$@"; do run $v; done
