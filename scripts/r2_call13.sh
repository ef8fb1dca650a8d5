R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc7_*
GEMM_BENCH_ONLY="input->gates NT" rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $O/pmc7_SQ -o pmc -- python $R/scripts/gemm_bench.py > $O/pmc7_SQ.log 2>&1
GEMM_BENCH_ONLY="input->gates NT" rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace -d $O/pmc7_LDS -o pmc -- python $R/scripts/gemm_bench.py > $O/pmc7_LDS.log 2>&1
cd $R; python scripts/rocpd_pmc_summary.py $(find $O/pmc7_SQ $O/pmc7_LDS -name "*.db") 2>&1 | grep -v rocclr | cut -c1-420
