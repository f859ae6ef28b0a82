cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q > $O/run34_tests.log 2>&1; echo "train tests rc=$?" > $O/run34_status.txt
rm -f $O/run34_ab.txt
for i in 1 2; do
DFH_TRAIN_GN_PRE=0 timeout 600 python bench.py --mode train --steps 6 --warmup 3 --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train gn_pre off', d['ms_per_step'])" >> $O/run34_ab.txt
timeout 600 python bench.py --mode train --steps 6 --warmup 3 --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train gn_pre on ', d['ms_per_step'])" >> $O/run34_ab.txt
done
tail -3 $O/run34_tests.log; cat $O/run34_status.txt; cat $O/run34_ab.txt
