cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
{
echo "== fused mlp tests (both forms)"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "mlp_fused" 2>&1 | grep -E "rel_l2|passed|failed|Error" | head -12
echo "== microbench form 1 / form 2"
FORM=1 timeout 300 python scripts/mlp_fused_microbench.py 2>&1 | tail -1
FORM=2 timeout 300 python scripts/mlp_fused_microbench.py 2>&1 | tail -1
FORM=2 timeout 300 python scripts/mlp_fused_microbench.py 16384 2>&1 | tail -1
} > $O/run7.txt 2>&1
cat $O/run7.txt
