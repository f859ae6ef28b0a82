cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', d['ms_per_step'], d['value'])"; }
{
echo "== op tests"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "mlp_fused or token_linear" 2>&1 | grep -E "rel_l2|passed|failed|Error|assert" | head -12
echo "== microbench"
FORM=2 timeout 300 python scripts/mlp_fused_microbench.py 2>&1 | tail -1
timeout 300 python scripts/token_linear_microbench.py 2>&1 | tail -3
echo "== census"
timeout 300 python scripts/census_step.py 2>&1 | tail -1
echo "== unet tests"
timeout 1200 python -m pytest tests/test_gpu_unet.py -q -x 2>&1 | tail -3
echo "== in-step A/B"
for i in 1 2 3; do
  DFH_TOKEN_LINEAR=0 DFH_MLP_FUSED=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-secondary 2>/dev/null | one "sampling gemm only       "
  DFH_TOKEN_LINEAR=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-secondary 2>/dev/null | one "sampling + fused MLP     "
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-secondary 2>/dev/null | one "sampling + token linears "
done
} > $O/run10.txt 2>&1
cat $O/run10.txt
