cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
DFH_ATTN_BWD_X32=1 timeout 900 python -m pytest tests/test_gpu_backward.py -x -q -k "attention_backward and 40-8" > $O/run30_tests.log 2>&1; echo "x32 tests rc=$?" > $O/run30_status.txt
rm -f $O/run30_ab.txt
for v in 0 1 2 3 0 1; do echo "== DFH_ATTN_BWD_X32=$v" >> $O/run30_ab.txt; DFH_ATTN_BWD_X32=$v timeout 300 python scripts/attn_bwd_microbench.py 2>&1 | grep "d= 40" >> $O/run30_ab.txt; done
tail -5 $O/run30_tests.log; cat $O/run30_status.txt; cat $O/run30_ab.txt
