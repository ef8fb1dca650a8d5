#!/bin/bash
# After `gpurun -- bash scripts/r5_final.sh`: copy the merged outputs of that call into profiles/ and stamp them with the commit and the
# source digest of the tree they were taken on (run on the SAME tree, before any further change under eesen_amd/csrc or include/).
C=$(git rev-parse --short HEAD)
cp gpurun_out/bench_r05.json profiles/r05_bench_line.json
for f in r05_kernel_stats.md r05_step_timeline.txt r05_pmc_fetch_write.md r05_pmc_sq.md r05_pmc_fetch_calibration.md; do cp gpurun_out/$f profiles/$f; done
cp gpurun_out/r5z/r05_s64_kernel_stats.md gpurun_out/r5z/r05_s64_step_timeline.txt profiles/
cp gpurun_out/r5z/multirank_persistent.json profiles/r05_multirank_persistent.json
cp gpurun_out/r5z/multirank_overlap.json profiles/r05_multirank_overlap.json
cp gpurun_out/r5z/gemm_accuracy.json profiles/r05_gemm_accuracy.json
cp gpurun_out/r5z/bf16_forward.json profiles/r05_bf16_forward.json
(echo "# closing GPU call of round 5 (scripts/r5_final.sh) on commit $C"; echo "## pytest tests -m gpu"; cat gpurun_out/r5z/test_gpu.log; echo "## smoke()"; cat gpurun_out/r5z/smoke.log
 echo "## bench.py --main-only --steps 20 --warmup 5, three more runs on the same box"; cat gpurun_out/r5z/headline_spread.log; echo "## scripts/soak.py cfg2 300 [64]"; cat gpurun_out/r5z/soak.log) > profiles/r05_final_call.log
python scripts/make_pmc_traffic.py r05 $C "EESEN_FWD_MID=0 (counter passes; rocprofv3 --pmc serialises kernels)" > /tmp/pmc_traffic.json && cp /tmp/pmc_traffic.json profiles/pmc_traffic.json
python scripts/split_parity.py gpurun_out/r5z/parity_fullsize.json $C
