cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_backward.py -x -q -k wgrad > $O/run15_tests.log 2>&1; echo "tests rc=$?" > $O/run15_status.txt
timeout 900 python scripts/wgrad_microbench.py 1 2 3 4 5 6 7 8 10 12 14 16 20 24 28 32 40 48 64 -2 -4 -8 0 > $O/run15_sweep.txt 2>&1
timeout 600 python bench.py --mode train --steps 6 --warmup 3 > $O/run15_train.json 2> $O/run15_train.err
DFH_WGRAD_PLAN=1 timeout 600 python bench.py --mode train --steps 6 --warmup 3 > $O/run15_train_oldplan.json 2> $O/run15_train_oldplan.err
timeout 600 python bench.py --mode train --steps 6 --warmup 3 > $O/run15_train2.json 2>> $O/run15_train.err
tail -3 $O/run15_tests.log; cat $O/run15_status.txt; cat $O/run15_sweep.txt; for f in $O/run15_train.json $O/run15_train_oldplan.json $O/run15_train2.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['value'])"; done
