#!/bin/bash
# Round 6, closing GPU calls, part B: counters and timelines of the wide configurations on the final tree (scripts/collect_profiles_wide.sh
# r06), soaks of the persistent kernels at --num-sequence 32 and 64, the kernel timeline of one --num-sequence 64 step.
mkdir -p gpurun_out/r6z; O=gpurun_out/r6z
export TMPDIR=/tmp
bash scripts/collect_profiles_wide.sh r06 > $O/collect_wide.log 2>&1; tail -60 $O/collect_wide.log
( timeout 200 python scripts/soak.py cfg2 300 2>&1 | tail -1; timeout 200 python scripts/soak.py cfg2 300 64 2>&1 | tail -1 ) > $O/soak.log 2>&1; cat $O/soak.log
R=$GRAFT_REPO_ROOT
cd /tmp && ( timeout 200 rocprofv3 --kernel-trace -d $R/$O/prof -o s64 -- python $R/bench.py --steps 3 --warmup 1 --main-only --S 64 > $R/$O/prof.log 2>&1 )
cd $R
DB=$(find $O/prof -name "*_results.db" | head -1)
if [ -n "$DB" ]; then
  python scripts/rocpd_summary.py $DB > $O/r06_s64_kernel_stats.md
  python scripts/timeline.py $DB > $O/r06_s64_step_timeline.txt 2>/dev/null
  head -12 $O/r06_s64_kernel_stats.md; tail -3 $O/r06_s64_step_timeline.txt
  rm -rf $O/prof
fi
# the self-describing N > 1 bench lines on record: the REAL librccl at the only world size a one-GPU box offers;
# two ranks each holding persistent grids on one GPU through the RCCL-shaped stand-in (2 x BiLSTM(256): co-resident); eight ranks
export HSA_ENABLE_IPC_MODE_LEGACY=0
FAKE=$PWD/tests/native/libfake_rccl.so
( MASTER_ADDR=127.0.0.1 MASTER_PORT=29547 timeout 300 python bench.py --gpus 1 --force-dist --steps 10 --warmup 3 --main-only 2>$O/bench_rccl_w1.err | grep '^{' ) > $O/r06_bench_real_rccl_world1.json
( EESEN_RCCL_LIBRARY=$FAKE FAKE_RCCL_QUIET=1 FAKE_RCCL_SHAPE=rccl EESEN_BENCH_SHARE_GPU=0 timeout 400 python bench.py --gpus 2 --steps 10 --warmup 3 --main-only --H 256 --layers 2 2>$O/bench_two.err | grep '^{' ) > $O/r06_bench_two_ranks_persistent.json
( EESEN_RCCL_LIBRARY=$FAKE FAKE_RCCL_QUIET=1 FAKE_RCCL_SHAPE=rccl FAKE_RCCL_BLOCKS=4 EESEN_BENCH_SHARE_GPU=0 timeout 900 python bench.py --gpus 8 --steps 3 --warmup 1 --main-only --H 64 --layers 2 --S 16 --T 200 2>$O/bench_eight.err | grep '^{' ) > $O/r06_bench_eight_ranks.json
python - <<'PY'
import json
for f in ("r06_bench_real_rccl_world1", "r06_bench_two_ranks_persistent", "r06_bench_eight_ranks"):
    try:
        d = json.load(open(f"gpurun_out/r6z/{f}.json")); c = d["config"]
        print(f, "n_gpus", d["n_gpus"], "ranks", c["ranks"], "stand_in", c["comm_stand_in"], "distinct", c["distinct_devices"], "world_seen", c["comm_world_seen"],
              "bit_identical", c["ranks_bit_identical"], "schedule", c["exchange_schedule"], "ms", round(d["ms_per_step"], 2))
    except Exception as e:
        print(f, "FAILED", e)
PY
# the recipe-shape leg at --num-sequence 10 / 32 under the arithmetic switches (where does S = 10 stand against round 5's 20.4 ms?)
for ns in 10 32; do
  for e in "EESEN_GEMM_MODE=half" "EESEN_GEMM_MODE=split" "EESEN_GEMM_MODE=split EESEN_FWD_F16=0" "EESEN_GEMM_MODE=half EESEN_FWD_F16=0"; do
    echo "recipe S=$ns $e: $(env $e timeout 200 python scripts/recipe_ab.py $ns 2>/dev/null | tail -1)" >> $O/recipe_ab.log
  done
done
cat $O/recipe_ab.log
