cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 600 python scripts/mlp_fused_diag.py > gpurun_out/r05/run4.txt 2>&1
cat gpurun_out/r05/run4.txt
