#!/bin/bash
# After `gpurun -- bash scripts/r6_final_a.sh` and `... r6_final_b.sh`: copy the merged outputs of those calls into profiles/ and stamp them
# with the commit and the source digest of the tree they were taken on (run on the SAME tree, before any further change under
# eesen_amd/csrc or include/).
C=$(git rev-parse --short HEAD)
Z=gpurun_out/r6z
cp gpurun_out/bench_r06.json profiles/r06_bench_line.json
for f in r06_kernel_stats.md r06_step_timeline.txt r06_pmc_fetch_write.md r06_pmc_sq.md r06_pmc_fetch_calibration.md; do cp gpurun_out/$f profiles/$f; done
for f in r06_s64_kernel_stats.md r06_s64_step_timeline.txt r06_plans.json; do [ -f $Z/$f ] && cp $Z/$f profiles/$f; done
for n in cfg4_f32 cfg4_bf16 cfg5; do for k in kernel_stats.md pmc_fetch_write.md pmc_sq.md step_timeline.txt; do [ -f gpurun_out/r06_${n}_$k ] && cp gpurun_out/r06_${n}_$k profiles/r06_${n}_$k; done; done
cp $Z/multirank_persistent.json profiles/r06_multirank_persistent.json
[ -f $Z/multirank_overlap_rccl.json ] && cp $Z/multirank_overlap_rccl.json profiles/r06_multirank_overlap_rccl.json
cp $Z/gemm_accuracy.json profiles/r06_gemm_accuracy.json
cp $Z/bf16_forward.json profiles/r06_bf16_forward.json
(echo "# closing GPU calls of round 6 (scripts/r6_final_a.sh, r6_final_b.sh) on commit $C"; echo "## pytest tests -m gpu"; cat $Z/test_gpu.log; echo "## smoke()"; cat $Z/smoke.log
 echo "## bench.py --main-only --steps 20 --warmup 5, three more runs on the same box"; cat $Z/headline_spread.log; echo "## scripts/soak.py cfg2 300 [64]"; cat $Z/soak.log 2>/dev/null) > profiles/r06_final_call.log
python scripts/make_pmc_traffic.py r06 $C "EESEN_FWD_MID=0 (counter passes; rocprofv3 --pmc serialises kernels)" > /tmp/pmc_traffic.json && cp /tmp/pmc_traffic.json profiles/pmc_traffic.json
python scripts/split_parity.py $Z/parity_fullsize.json $C
python tools/front_page.py > /dev/null
for f in r06_bench_real_rccl_world1.json r06_bench_two_ranks_persistent.json r06_bench_eight_ranks.json; do [ -s $Z/$f ] && cp $Z/$f profiles/$f; done
