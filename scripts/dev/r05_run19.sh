cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_backward.py -x -q -k attention_backward > $O/run19_tests.log 2>&1; echo "attn bwd tests rc=$?" > $O/run19_status.txt
echo "== this tree" > $O/run19_ab.txt
timeout 300 python scripts/attn_bwd_microbench.py >> $O/run19_ab.txt 2>&1
echo "== previous commit (DFH_LIB=gpurun_ab/libdifashion_hip_prev.so)" >> $O/run19_ab.txt
DFH_LIB=gpurun_ab/libdifashion_hip_prev.so timeout 300 python scripts/attn_bwd_microbench.py >> $O/run19_ab.txt 2>&1
for i in 1 2; do
timeout 600 python bench.py --mode train --steps 6 --warmup 3 --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train this tree', d['ms_per_step'])" >> $O/run19_ab.txt
DFH_LIB=gpurun_ab/libdifashion_hip_prev.so timeout 600 python bench.py --mode train --steps 6 --warmup 3 --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train previous', d['ms_per_step'])" >> $O/run19_ab.txt
done
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q > $O/run19_train_tests.log 2>&1; echo "train tests rc=$?" >> $O/run19_status.txt
tail -3 $O/run19_tests.log; cat $O/run19_status.txt; cat $O/run19_ab.txt; tail -3 $O/run19_train_tests.log
