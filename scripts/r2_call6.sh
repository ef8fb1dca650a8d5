R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
for v in base pf2 pf2w3; do
  L=""; [ $v != base ] && L="$R/eesen_amd/lib/variants/libeesen_hip_$v.so"
  echo "== $v"; EESEN_HIP_LIBRARY=$L python scripts/gemm_bench.py 2>&1 | grep "bf16-split" | grep -v "L1 NT\|affine\|one tile"
  EESEN_HIP_LIBRARY=$L timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x 2>&1 | tail -2
done
