cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -s -k "fp8_sampler" 2>&1 | grep -E "fp8:|passed|failed" > gpurun_out/r05/run36_fp8_err.txt
cat gpurun_out/r05/run36_fp8_err.txt
