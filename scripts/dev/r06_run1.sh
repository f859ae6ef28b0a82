# round 6, GPU run 1: why is the side-stream test 243 s?  + the re-built library under the touched tests + a baseline bench line
O=gpurun_out/r06_run1; mkdir -p $O
for e in "DFH_TRAIN_SIDE=0" "DFH_TRAIN_SIDE_MIN_FLOP=0"; do
  ( export $e DFH_WORKER_TIMING=1; /usr/bin/time -v python tests/train_side_worker.py /tmp/w.pt ) > $O/side_worker_$e.log 2>&1
done
grep -h "worker\|Elapsed" $O/side_worker_*.log
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py tests/test_gpu_pipeline.py -m gpu -q -x -k "pndm or gstat or statistics or dup_tail or scheduler or golden or teacher" --durations=8 2>&1 | tail -15 > $O/touched_tests.log; tail -4 $O/touched_tests.log
python bench.py > $O/bench_base.json 2> $O/bench_base.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_run1/bench_base.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], {k:v.get('ms_per_step') for k,v in d.get('secondary_configs',{}).items()})
PY
