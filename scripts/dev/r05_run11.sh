cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
{
echo "== op tests"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "mlp_fused or token_linear" 2>&1 | grep -E "rel_l2|passed|failed|Error|assert" | head -12
echo "== microbench"
FORM=2 timeout 300 python scripts/mlp_fused_microbench.py 2>&1 | tail -1
FORM=2 timeout 300 python scripts/mlp_fused_microbench.py 16384 2>&1 | tail -1
timeout 300 python scripts/token_linear_microbench.py 2>&1 | tail -3
} > $O/run11.txt 2>&1
cat $O/run11.txt
