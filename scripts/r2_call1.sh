# round 2, GPU call 1: full GPU suite (incl. full-length reference parity), bench A/B of the XCD-aware GEMM map, native comm path, PMC traffic
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest_r2a.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_r2a.log
python bench.py --steps 10 --warmup 3 > $O/bench_r2a.json 2> $O/bench_r2a.err; echo "bench rc=$?"
EESEN_GEMM_XCD=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_r2a_noxcd.json 2> $O/bench_r2a_noxcd.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --force-dist > $O/bench_r2a_dist.json 2> $O/bench_r2a_dist.err; echo "dist rc=$?"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --force-dist --comm bulk > $O/bench_r2a_bulk.json 2> $O/bench_r2a_bulk.err
for f in bench_r2a bench_r2a_noxcd bench_r2a_dist bench_r2a_bulk; do python - <<PY
import json
try:
    d=json.load(open("$O/$f.json")); p=d["phase_ms_per_step"]
    print("$f", round(d["ms_per_step"],2), {k: round(v,2) for k,v in p.items()}, d["roofline"].get("gate_gemm_standalone",{}).get("achieved"))
except Exception as e: print("$f", "FAILED", e)
PY
done
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc5_$c
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc5_$c -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc5_$c.log 2>&1
done
cd $R; python scripts/rocpd_pmc_summary.py $O/pmc5_FETCH_SIZE/*/pmc_results.db $O/pmc5_WRITE_SIZE/*/pmc_results.db > $O/r02a_pmc_fetch_write.md 2>$O/pmc_sum.err || python scripts/rocpd_pmc_summary.py $(find $O/pmc5_FETCH_SIZE $O/pmc5_WRITE_SIZE -name "*.db") > $O/r02a_pmc_fetch_write.md
head -20 $O/r02a_pmc_fetch_write.md
