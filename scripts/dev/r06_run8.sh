# round 6, GPU run 8: CLIP end-to-end test, smoke(), bench --mode clip (both shapes), default bench line
O=gpurun_out/r06_run8; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_clip.py -m gpu -q -x 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
python bench.py --mode clip --steps 10 --warmup 2 > $O/bench_clip.json 2> $O/bench_clip.err; cat $O/bench_clip.json | cut -c1-1500
python bench.py --mode clip --config sd2base --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_clip_sd2.json 2>/dev/null; cat $O/bench_clip_sd2.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/clip_trace -o clip -- python3 $GRAFT_REPO_ROOT/bench.py --mode clip --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
head -12 $O/clip_trace/clip_kernel_stats.csv | cut -c1-200
