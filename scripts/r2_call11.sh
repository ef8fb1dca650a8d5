R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_gemm.py "tests/test_gpu_parity.py::test_gemm" -q 2>&1 | tail -4
python scripts/gemm_bench.py 2>&1 | grep "bf16-split" | grep -v "L1 NT\|affine\|one tile"
EESEN_GEMM_BIG=0 python scripts/gemm_bench.py 2>&1 | grep "bf16-split" | grep "input->gates NT\|in_diff"
for cfgx in cfg2 cfg4; do
for big in 1 0; do
  EESEN_GEMM_BIG=$big python bench.py --config $cfgx --steps 8 --warmup 3 --main-only > $O/bench_r2i_${cfgx}_$big.json 2> $O/bench_r2i_${cfgx}_$big.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_r2i_${cfgx}_$big.json")); p=d["phase_ms_per_step"]
    print("$cfgx big=$big", round(d["ms_per_step"],2), {k: round(v,2) for k,v in p.items() if not k.startswith("ctc")})
except Exception as e: print("$cfgx big=$big FAILED", e)
PY
done; done
