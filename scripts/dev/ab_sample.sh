# same-box A/B of the sampling step: DFH_LIB = previous build vs the tree's build
for i in 1 2 3; do
  DFH_LIB=$GRAFT_REPO_ROOT/gpurun_ab/libdifashion_hip_prev.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('prev', d['ms_per_step'])"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new ', d['ms_per_step'])"
done
