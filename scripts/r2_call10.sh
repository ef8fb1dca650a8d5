R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_gemm.py -q 2>&1 | tail -4
cat $O/bf16_forward.json | head -12
for fp in f32 bf16; do
  python bench.py --config cfg4 --steps 5 --warmup 2 --no-cpu-baseline --forward-precision $fp > $O/bench_r2h_cfg4_$fp.json 2> $O/bench_r2h_cfg4_$fp.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_r2h_cfg4_$fp.json")); p=d["phase_ms_per_step"]
    print("cfg4 $fp", round(d["ms_per_step"],2), round(d["value"]), d["dtype"], {k: round(v,2) for k,v in p.items() if not k.startswith("ctc")}, d["roofline"]["whole_step"])
except Exception as e: print("cfg4 $fp FAILED", e)
PY
done
