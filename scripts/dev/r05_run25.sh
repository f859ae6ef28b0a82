cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 300 python scripts/hbm_write_probe.py > gpurun_out/r05/run25_hbm.txt 2>&1
cat gpurun_out/r05/run25_hbm.txt
