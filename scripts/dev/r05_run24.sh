cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q -k "dup_tail or fp8" > $O/run24_tests.log 2>&1; echo "unet dup/fp8 tests rc=$?" > $O/run24_status.txt
timeout 1500 python -m pytest tests/test_gpu_pipeline.py -x -q >> $O/run24_tests.log 2>&1; echo "pipeline tests rc=$?" >> $O/run24_status.txt
rm -f $O/run24_ab.txt
for i in 1 2; do
DFH_CFG_DEDUP=0 timeout 300 python bench.py --dtype fp8 --no-secondary --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp8 dedup off', d['ms_per_step'])" >> $O/run24_ab.txt
timeout 300 python bench.py --dtype fp8 --no-secondary --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp8 dedup on ', d['ms_per_step'])" >> $O/run24_ab.txt
timeout 300 python bench.py --no-secondary --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16         ', d['ms_per_step'])" >> $O/run24_ab.txt
done
timeout 300 python scripts/gemm_write_probe.py > $O/run24_write_probe.txt 2>&1
tail -4 $O/run24_tests.log; cat $O/run24_status.txt; cat $O/run24_ab.txt; cat $O/run24_write_probe.txt
