cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', d['ms_per_step'], d['value'])"; }
{
for i in 1 2 3; do
  for f in 0 1 2; do
  DFH_MLP_FUSED=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-secondary 2>/dev/null | one "sampling DFH_MLP_FUSED=$f"
  done
done
timeout 900 python -m pytest tests/test_gpu_unet.py -q -k "batch16 or tiny" 2>&1 | tail -3
} > $O/run8.txt 2>&1
cat $O/run8.txt
