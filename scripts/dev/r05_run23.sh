cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q -k "dup_tail or run_cache or small" > $O/run23_tests.log 2>&1; echo "unet dup tests rc=$?" > $O/run23_status.txt
timeout 1500 python -m pytest tests/test_gpu_pipeline.py -x -q >> $O/run23_tests.log 2>&1; echo "pipeline tests rc=$?" >> $O/run23_status.txt
rm -f $O/run23_ab.txt
for i in 1 2; do
DFH_CFG_DEDUP=0 timeout 300 python bench.py --no-secondary --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dedup off', d['ms_per_step'])" >> $O/run23_ab.txt
timeout 300 python bench.py --no-secondary --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dedup on ', d['ms_per_step'])" >> $O/run23_ab.txt
done
tail -5 $O/run23_tests.log; cat $O/run23_status.txt; cat $O/run23_ab.txt
