# round 5, GPU run 1: full -m gpu suite, same-box A/B (round-4 library through DFH_LIB vs this tree), the default bench line with its secondary legs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
O=gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -x -q > $O/run1_tests.log 2>&1; echo "tests rc=$?" > $O/run1_status.txt
one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', d['ms_per_step'], d['value'])"; }
{
for i in 1 2 3; do
  DFH_LIB=$GRAFT_REPO_ROOT/gpurun_ab/libdifashion_hip_r04.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-secondary 2>/dev/null | one "sampling r04-lib "
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-secondary 2>/dev/null | one "sampling HEAD    "
done
for i in 1 2; do
  DFH_LIB=$GRAFT_REPO_ROOT/gpurun_ab/libdifashion_hip_r04.so python bench.py --mode train --steps 6 --warmup 2 --no-profile --no-cpu-baseline 2>/dev/null | one "training r04-lib "
  python bench.py --mode train --steps 6 --warmup 2 --no-profile --no-cpu-baseline 2>/dev/null | one "training HEAD    "
done
for i in 1 2; do
  DFH_LIB=$GRAFT_REPO_ROOT/gpurun_ab/libdifashion_hip_r04.so python bench.py --steps 20 --warmup 5 --dtype fp8 --no-cpu-baseline --no-profile --no-secondary 2>/dev/null | one "fp8      r04-lib "
  python bench.py --steps 20 --warmup 5 --dtype fp8 --no-cpu-baseline --no-profile --no-secondary 2>/dev/null | one "fp8      HEAD    "
done
} > $O/run1_ab.txt 2>&1
python bench.py > $O/run1_bench_default.json 2> $O/run1_bench_default.err; echo "bench rc=$?" >> $O/run1_status.txt
python scripts/attn_sd2_microbench.py > $O/run1_attn_sd2.txt 2>&1
DFH_PROF_TABLE=$O/run1_launch_table.txt python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1
tail -3 $O/run1_tests.log; cat $O/run1_status.txt $O/run1_ab.txt; python -c "
import json; d=json.loads([l for l in open('$O/run1_bench_default.json') if l.startswith('{')][0]); print(d['ms_per_step'], d['value']); s=d.get('secondary_configs',{}); print({k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in s.items()}); print(s.get('seconds_spent'))"
