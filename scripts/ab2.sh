# repeated full-step timing: bash scripts/ab2.sh REPS name[:ENV=..,ENV=..] ...
REPS=$1; shift
for r in $(seq $REPS); do
  for v in "$@"; do
    name=${v%%:*}; envs=""; [ "$v" != "$name" ] && envs=$(echo "${v#*:}" | tr ',' ' ')
    L=""; [ "$name" != base ] && L="eesen_amd/lib/variants/libeesen_hip_$name.so"
    env EESEN_HIP_LIBRARY=$L $envs python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2))"
  done
done
