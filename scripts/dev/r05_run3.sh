cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
{
for d in 0 1; do
echo "== fused mlp test DFH_MLP_DBG=$d"
DFH_MLP_DBG=$d timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "mlp_fused" 2>&1 | grep -E "rel_l2|passed|failed|Error" | head -12
DFH_MLP_DBG=$d timeout 300 python scripts/mlp_fused_microbench.py 2>&1 | tail -1
done
} > $O/run3.txt 2>&1
cat $O/run3.txt
