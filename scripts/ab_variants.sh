#!/bin/bash
# A/B on one box: the regular build against variant libraries (scripts/build_variant.py), bench main leg only.
# usage: ab_variants.sh "<bench args>" name1 name2 ...
args="$1"; shift
mkdir -p gpurun_out
for round in 1 2; do
  for v in base "$@"; do
    if [ "$v" = base ]; then unset EESEN_HIP_LIBRARY; else export EESEN_HIP_LIBRARY=$PWD/eesen_amd/lib/variants/libeesen_hip_$v.so; fi
    python bench.py --main-only $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']
print('$v', round(d['ms_per_step'],2), 'fwd', round(p['recurrence_fwd'],2), 'bwd', round(p['recurrence_bwd'],2))"
  done
done
