# round 5, GPU run 2: the two in-wave-pipelined kernels -- attention with four query blocks per wave (DFH_ATTN_QB4=1) and the fused MLP
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
{
echo "== attention tests, DFH_ATTN_QB4=1"
DFH_ATTN_QB4=1 timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" 2>&1 | tail -5
echo "== attention microbench default"
timeout 300 python scripts/attn_microbench.py 2>&1 | head -3
echo "== attention microbench DFH_ATTN_QB4=1"
DFH_ATTN_QB4=1 timeout 300 python scripts/attn_microbench.py 2>&1 | head -3
echo "== fused mlp test"
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "mlp_fused" 2>&1 | tail -15
echo "== fused mlp microbench"
timeout 300 python scripts/mlp_fused_microbench.py 2>&1 | tail -3
timeout 300 python scripts/mlp_fused_microbench.py 16384 2>&1 | tail -1
} > $O/run2.txt 2>&1
cat $O/run2.txt
