cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', d['ms_per_step'], d['value'])"; }
{
for i in 1 2 3; do
  DFH_MLP_FUSED=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-secondary 2>/dev/null | one "sampling two-launch MLP"
  DFH_MLP_FUSED=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-secondary 2>/dev/null | one "sampling fused MLP     "
done
DFH_MLP_FUSED=1 DFH_PROF_TABLE=$O/run6_launch_table_fused.txt python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print({k:v['ms_per_step'] for k,v in d['kernel_classes'].items()})"
DFH_MLP_FUSED=0 DFH_PROF_TABLE=$O/run6_launch_table_two.txt python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print({k:v['ms_per_step'] for k,v in d['kernel_classes'].items()})"
timeout 600 python -m pytest tests/test_gpu_unet.py -q -k "batch16" 2>&1 | tail -3
} > $O/run6.txt 2>&1
cat $O/run6.txt
