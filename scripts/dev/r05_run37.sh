cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests/test_gpu_unet.py -x -q -s -k "batch16_matches_oracle or sd2base_full_size" 2>&1 | grep -E "sd15 B=16|sd2|fp8|passed|failed|e-0" > gpurun_out/r05/run37_parity.txt
cat gpurun_out/r05/run37_parity.txt
