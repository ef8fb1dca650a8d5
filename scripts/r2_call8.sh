R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_r2f.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_r2f.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_r2f.json 2> $O/bench_r2f.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_r2f.json")); p=d["phase_ms_per_step"]
print("cfg2", round(d["ms_per_step"],2), {k: round(v,3) for k,v in p.items()}); print(json.dumps(d["roofline"]["ctc"]))
PY
timeout 900 python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_r2f_cfg5.json 2> $O/bench_r2f_cfg5.err
python - <<PY
import json
d=json.load(open("$O/bench_r2f_cfg5.json")); print("cfg5", round(d["ms_per_step"],1)); print(json.dumps(d["roofline"]["ctc"]))
PY
