R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
for v in base pf2 w3 pf2w3; do
  L=""; [ $v != base ] && L="$R/eesen_amd/lib/variants/libeesen_hip_$v.so"
  echo "== $v"; EESEN_HIP_LIBRARY=$L python scripts/gemm_bench.py 2>&1 | grep "bf16-split" | grep -v "L1 NT\|affine\|one tile"
done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc6_*
GEMM_BENCH_ONLY="input->gates NT" rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $O/pmc6_SQ -o pmc -- python $R/scripts/gemm_bench.py > $O/pmc6_SQ.log 2>&1
GEMM_BENCH_ONLY="input->gates NT" rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace -d $O/pmc6_LDS -o pmc -- python $R/scripts/gemm_bench.py > $O/pmc6_LDS.log 2>&1
cd $R; python scripts/rocpd_pmc_summary.py $(find $O/pmc6_SQ $O/pmc6_LDS -name "*.db") 2>&1 | grep -v rocclr | cut -c1-400
