R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_gemm.py "tests/test_gpu_parity.py::test_gemm" -q > $O/pytest_r2c.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_r2c.log
python scripts/gemm_bench.py 2>&1 | tee $O/gemm_bench_r2c.txt
for m in f32 split; do
  EESEN_GEMM_MODE=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_r2c_$m.json 2> $O/bench_r2c_$m.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_r2c_$m.json")); p=d["phase_ms_per_step"]
    print("$m", round(d["ms_per_step"],2), {k: round(v,2) for k,v in p.items()}, d["roofline"].get("gate_gemm_standalone",{}).get("achieved"))
except Exception as e: print("$m", "FAILED", e)
PY
done
