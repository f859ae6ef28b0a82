cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "groupnorm_fold" > $O/run18_tests.log 2>&1; echo "fold op tests rc=$?" > $O/run18_status.txt
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q >> $O/run18_tests.log 2>&1; echo "unet tests rc=$?" >> $O/run18_status.txt
for i in 1 2; do
DFH_GN_FOLD=0 timeout 300 python bench.py --no-secondary --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fold off', d['ms_per_step'])" >> $O/run18_ab.txt
DFH_GN_FOLD=320 timeout 300 python bench.py --no-secondary --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fold 320', d['ms_per_step'])" >> $O/run18_ab.txt
DFH_GN_FOLD=640 timeout 300 python bench.py --no-secondary --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fold 640', d['ms_per_step'])" >> $O/run18_ab.txt
DFH_GN_FOLD=1280 timeout 300 python bench.py --no-secondary --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fold 1280', d['ms_per_step'])" >> $O/run18_ab.txt
done
tail -5 $O/run18_tests.log; cat $O/run18_status.txt; cat $O/run18_ab.txt
