cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
rm -f $O/run21_ab.txt
for i in 1 2; do
timeout 600 python bench.py --mode train --steps 6 --warmup 3 --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train this tree', d['ms_per_step'])" >> $O/run21_ab.txt
DFH_LIB=gpurun_ab/libdifashion_hip_prev.so timeout 600 python bench.py --mode train --steps 6 --warmup 3 --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train previous', d['ms_per_step'])" >> $O/run21_ab.txt
done
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py -x -q > $O/run21_train_tests.log 2>&1; echo "train+backward tests rc=$?" > $O/run21_status.txt
cat $O/run21_status.txt; cat $O/run21_ab.txt; tail -3 $O/run21_train_tests.log
