R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_gemm.py -q 2>&1 | tail -3
python scripts/gemm_bench.py 2>&1 | grep "bf16-split" | grep -v "L1 NT\|affine\|one tile"
for cfgx in cfg2 cfg4; do
  python bench.py --config $cfgx --steps 8 --warmup 3 --main-only > $O/bench_r2j_${cfgx}.json 2> $O/bench_r2j_${cfgx}.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_r2j_${cfgx}.json")); p=d["phase_ms_per_step"]
    print("$cfgx", round(d["ms_per_step"],2), {k: round(v,2) for k,v in p.items() if not k.startswith("ctc")})
except Exception as e: print("$cfgx FAILED", e)
PY
done
