cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 120 scripts/probes/grid_barrier > gpurun_out/r05/run27_grid_barrier.txt 2>&1
cat gpurun_out/r05/run27_grid_barrier.txt
timeout 300 python scripts/gemm_scaling_probe.py 2>&1 | head -12
