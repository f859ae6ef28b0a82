cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash scripts/profile_train.sh r05 > gpurun_out/r05/run16_prof.txt 2>&1
python scripts/train_timeline.py gpurun_out/prof_train_r05/trace > gpurun_out/r05/run16_timeline.txt 2>&1
rm -rf gpurun_out/prof_train_r05/trace
cat gpurun_out/r05/run16_timeline.txt; tail -3 gpurun_out/r05/run16_prof.txt
