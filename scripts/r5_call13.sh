#!/bin/bash
# Round 5, GPU call 13: bench.py --gpus 2 with PERSISTENT kernels in both ranks on one GPU (EESEN_GPU_SHARE=2, a co-resident shape:
# 2 x BiLSTM of 256 cells, S = 32 per rank) through the stand-in collective -- the bench line with roofline.exchange of the product's
# N > 1 configuration ("Done").  Plumbing and spans, not an xGMI measurement.
mkdir -p gpurun_out/r5m; O=gpurun_out/r5m
export TMPDIR=/tmp
unset RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR MASTER_PORT
for defer in 0 1; do
  ( EESEN_RCCL_LIBRARY=$PWD/tests/native/libfake_rccl.so FAKE_RCCL_QUIET=1 EESEN_BENCH_SHARE_GPU=0 EESEN_GPU_SHARE=2 EESEN_COMM_DEFER=$defer HSA_ENABLE_IPC_MODE_LEGACY=0 \
    timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --main-only --H 256 --layers 2 2>$O/err_$defer.log | tail -1 ) > $O/bench_two_ranks_persistent_defer$defer.json
  python - <<P
import json
d = json.load(open("$O/bench_two_ranks_persistent_defer$defer.json"))
ex = d["roofline"].get("exchange", {})
print("defer=$defer", "n_gpus", d["n_gpus"], round(d["ms_per_step"], 3), "ms", round(d["value"]), "fps; exchange ms", round(ex.get("ms_per_step", 0), 3), "exposed", round(ex.get("exposed_ms_per_step", 0), 3), [round(b["MB"], 2) for b in ex.get("buckets", [])], d["config"].get("exchange_schedule", "")[:20])
P
  grep -a -h "WARNING\|persistent path failed" $O/err_$defer.log | head -3
done
