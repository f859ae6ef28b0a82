cd $GRAFT_REPO_ROOT
for i in 1 2; do
(cd gpurun_ab/r03 && DFH_PROF_TABLE=$GRAFT_REPO_ROOT/gpurun_out/lt_r03_$i.txt python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > $GRAFT_REPO_ROOT/gpurun_out/cls_r03_$i.json)
DFH_PROF_TABLE=$GRAFT_REPO_ROOT/gpurun_out/lt_head_$i.txt python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/cls_head_$i.json
done
