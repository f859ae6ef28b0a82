cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_backward.py -x -q -k attention_backward > $O/run20_tests.log 2>&1; echo "attn bwd tests rc=$?" > $O/run20_status.txt
echo "== this tree" > $O/run20_ab.txt
timeout 300 python scripts/attn_bwd_microbench.py >> $O/run20_ab.txt 2>&1
echo "== previous commit (DFH_LIB=gpurun_ab/libdifashion_hip_prev.so)" >> $O/run20_ab.txt
DFH_LIB=gpurun_ab/libdifashion_hip_prev.so timeout 300 python scripts/attn_bwd_microbench.py >> $O/run20_ab.txt 2>&1
tail -3 $O/run20_tests.log; cat $O/run20_status.txt; cat $O/run20_ab.txt
