cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_sample_tl
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o tl -- python3 $R/bench.py --steps 4 --warmup 2 --no-secondary --no-profile > $O/trace.log 2>&1
cd $R
python scripts/train_timeline.py $O/trace cfg_step_kernel > gpurun_out/r05/run17_sample_timeline.txt 2>&1
rm -rf $O/trace
cat gpurun_out/r05/run17_sample_timeline.txt; tail -2 $O/trace.log
