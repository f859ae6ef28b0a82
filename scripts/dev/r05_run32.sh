cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "groupnorm_statistics or gstat" > $O/run32_tests.log 2>&1; echo "gstat op tests rc=$?" > $O/run32_status.txt
timeout 1500 python -m pytest tests/test_gpu_unet.py -x -q >> $O/run32_tests.log 2>&1; echo "unet tests rc=$?" >> $O/run32_status.txt
python scripts/census_step.py > $O/run32_census.txt 2>&1
DFH_GSTAT128=0 python scripts/census_step.py > $O/run32_census_off.txt 2>&1
rm -f $O/run32_ab.txt
for i in 1 2 3; do
DFH_GSTAT128=0 timeout 300 python bench.py --no-secondary --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gstat128 off', d['ms_per_step'])" >> $O/run32_ab.txt
timeout 300 python bench.py --no-secondary --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gstat128 on ', d['ms_per_step'])" >> $O/run32_ab.txt
done
tail -4 $O/run32_tests.log; cat $O/run32_status.txt; cat $O/run32_ab.txt; grep -i "gn_\|gstat" $O/run32_census.txt | head; echo ---; grep -i "gn_\|gstat" $O/run32_census_off.txt | head
