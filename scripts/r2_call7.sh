R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_r2e.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_r2e.log
python bench.py --steps 10 --warmup 3 > $O/bench_r2e.json 2> $O/bench_r2e.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_r2e.json")); p=d["phase_ms_per_step"]
print("cfg2", round(d["ms_per_step"],2), {k: round(v,2) for k,v in p.items() if not k.startswith("ctc")}, d["config"].get("f32_mfma_gemm_only"), d["cpu_baseline"].get("value"))
PY
timeout 900 python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_r2e_cfg5.json 2> $O/bench_r2e_cfg5.err; echo "cfg5 rc=$?"; tail -3 $O/bench_r2e_cfg5.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench_r2e_cfg5.json")); p=d["phase_ms_per_step"]
    print("cfg5", round(d["ms_per_step"],2), d["value"], {k: round(v,2) for k,v in p.items()})
except Exception as e: print("cfg5 FAILED", e)
PY
