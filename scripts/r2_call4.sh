R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
run() { # name, env...
  n=$1; shift
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_r2d_$n.json 2> $O/bench_r2d_$n.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_r2d_$n.json")); p=d["phase_ms_per_step"]
    print("$n", round(d["ms_per_step"],2), {k: round(v,2) for k,v in p.items() if not k.startswith("ctc")})
except Exception as e: print("$n", "FAILED", e)
PY
}
run split_default EESEN_GEMM_MODE=split
run split_nogate EESEN_GEMM_MODE=split EESEN_GATE_FWD=0
run split_nooverlap EESEN_GEMM_MODE=split EESEN_OVERLAP=0
run split_serial EESEN_GEMM_MODE=split EESEN_OVERLAP=0 EESEN_GATE_FWD=0
run split_side48 EESEN_GEMM_MODE=split EESEN_SIDE_LDS_KB=48
run split_nogate_side48 EESEN_GEMM_MODE=split EESEN_GATE_FWD=0 EESEN_SIDE_LDS_KB=48
run split_xcd0 EESEN_GEMM_MODE=split EESEN_GEMM_XCD=0
run f32_serial EESEN_GEMM_MODE=f32 EESEN_OVERLAP=0 EESEN_GATE_FWD=0
