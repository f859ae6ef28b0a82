cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 600 python scripts/census_step.py 1 > gpurun_out/r05/run35_census16.txt 2>&1
timeout 600 python scripts/census_step.py 4 > gpurun_out/r05/run35_census64.txt 2>&1
tail -1 gpurun_out/r05/run35_census16.txt; tail -1 gpurun_out/r05/run35_census64.txt
DFH_PROF_TABLE=gpurun_out/r05/run35_launch_table_batch64.txt timeout 600 python bench.py --outfits-per-gpu 4 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
head -12 gpurun_out/r05/run35_launch_table_batch64.txt
