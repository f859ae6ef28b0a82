cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 600 python scripts/gemm_scaling_probe.py > gpurun_out/r05/run22_scaling.txt 2>&1
cat gpurun_out/r05/run22_scaling.txt
