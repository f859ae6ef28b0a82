# round 6, GPU run 7: same-box A/B of DFH_REFILL_MID (ring refill between the two halves of a k-step instead of right behind the barrier)
O=gpurun_out/r06_run7; mkdir -p $O
one() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-secondary 2>/dev/null | one "sampling base      "
  DFH_LIB=$GRAFT_REPO_ROOT/gpurun_ab/refill_mid/libdifashion_hip.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-secondary 2>/dev/null | one "sampling refill-mid"
done | tee $O/refill_mid_ab.txt
for i in 1 2; do
  python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | one "training base      "
  DFH_LIB=$GRAFT_REPO_ROOT/gpurun_ab/refill_mid/libdifashion_hip.so python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | one "training refill-mid"
done | tee -a $O/refill_mid_ab.txt
DFH_LIB=$GRAFT_REPO_ROOT/gpurun_ab/refill_mid/libdifashion_hip.so timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "gemm or conv" 2>&1 | tail -3 | tee -a $O/refill_mid_ab.txt
