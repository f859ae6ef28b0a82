cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', d['ms_per_step'], d['value'], d.get('loss'))"; }
{
echo "== wgrad / backward tests"
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py -q -x 2>&1 | tail -3
echo "== wgrad microbench: r04 library"
DFH_LIB=$GRAFT_REPO_ROOT/gpurun_ab/libdifashion_hip_r04.so timeout 600 python scripts/wgrad_microbench.py 2>&1 | tail -18
echo "== wgrad microbench: this tree (register-staged)"
timeout 600 python scripts/wgrad_microbench.py 2>&1 | tail -18
echo "== training step A/B"
for i in 1 2; do
  DFH_LIB=$GRAFT_REPO_ROOT/gpurun_ab/libdifashion_hip_r04.so python bench.py --mode train --steps 6 --warmup 2 --no-profile --no-cpu-baseline 2>/dev/null | one "training r04-lib "
  python bench.py --mode train --steps 6 --warmup 2 --no-profile --no-cpu-baseline 2>/dev/null | one "training HEAD    "
done
} > $O/run12.txt 2>&1
cat $O/run12.txt
