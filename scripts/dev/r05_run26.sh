cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 bash scripts/pmc_gemm.sh lin320k320 > gpurun_out/r05/run26_pmc_lin320k320.txt 2>&1
timeout 900 bash scripts/pmc_gemm.sh lin960k320 > gpurun_out/r05/run26_pmc_lin960k320.txt 2>&1
rm -rf gpurun_out/pmc_lin320k320 gpurun_out/pmc_lin960k320
timeout 600 python -m pytest tests/test_gpu_unet.py -x -q -k "dup_tail" > gpurun_out/r05/run26_tests.log 2>&1; echo "dup tests rc=$?"
cat gpurun_out/r05/run26_pmc_lin320k320.txt gpurun_out/r05/run26_pmc_lin960k320.txt; tail -3 gpurun_out/r05/run26_tests.log
