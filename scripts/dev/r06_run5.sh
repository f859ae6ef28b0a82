# round 6, GPU run 5: the whole -m gpu suite with durations (target <= 450 s), clock / power traces, rocprofv3 passes of the round
O=gpurun_out/r06_run5; mkdir -p $O
( time python -m pytest tests -m gpu -q --durations=25 ) > $O/suite.log 2>&1; tail -32 $O/suite.log
hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_rate2.hip -o /tmp/mfma_rate2 2>/dev/null
python scripts/clock_trace.py $O/clock_trace_mfma_only.txt -- /tmp/mfma_rate2 > $O/mfma_rate2.out 2>&1; head -12 $O/clock_trace_mfma_only.txt
python scripts/clock_trace.py $O/clock_trace_sampling.txt -- python bench.py --steps 300 --warmup 5 --no-profile --no-cpu-baseline --no-secondary > $O/bench_300.out 2>&1; head -12 $O/clock_trace_sampling.txt; tail -1 $O/bench_300.out | cut -c1-200
