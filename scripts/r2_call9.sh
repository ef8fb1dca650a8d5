R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_r2g.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_r2g.log
python __graft_entry__.py --smoke 2>&1 | tail -2
