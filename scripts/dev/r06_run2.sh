# round 6, GPU run 2: persistent token-linear kernel -- parity vs the tile kernel, microbench, in-step A/B; side-stream worker timing
O=gpurun_out/r06_run2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "persistent" 2>&1 | tail -25 > $O/persist_test.log; tail -12 $O/persist_test.log
timeout 600 python scripts/gemm_persist_microbench.py > $O/persist_microbench.txt 2>&1; grep -v amdgpu.ids $O/persist_microbench.txt | tail -50
for e in "DFH_TRAIN_SIDE=0" "DFH_TRAIN_SIDE_MIN_FLOP=0"; do
  ( export $e DFH_WORKER_TIMING=1; time python tests/train_side_worker.py /tmp/w.pt ) > "$O/side_worker_$e.log" 2>&1
  grep -h "worker\|real" "$O/side_worker_$e.log"
done
for i in 1 2; do
  for m in 0 1; do
    DFH_PERSIST=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('DFH_PERSIST=$m', d['ms_per_step'])"
  done
done | tee $O/instep_ab.txt
