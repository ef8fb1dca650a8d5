# same-box comparison of built worktrees: bash scripts/ab_old.sh REPS dir1 dir2 ... ('.' = this tree); ENVS="A=1 B=2" passes env
REPS=$1; shift
for r in $(seq $REPS); do
  for d in "$@"; do
    (cd $d && env $ENVS python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('$d', round(d['ms_per_step'],2), round(p['recurrence_fwd'],2), round(p['recurrence_bwd'],2), round(p['grad_gemm'],2))")
  done
done
