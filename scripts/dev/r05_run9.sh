cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -x -q > $O/run9_tests.log 2>&1; echo "tests rc=$?" > $O/run9_status.txt
FORM=2 timeout 300 python scripts/mlp_fused_microbench.py > $O/run9_mlp2.txt 2>&1
tail -4 $O/run9_tests.log; cat $O/run9_status.txt; tail -1 $O/run9_mlp2.txt
