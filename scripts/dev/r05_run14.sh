cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_backward.py -x -q -k wgrad > $O/run14_tests.log 2>&1; echo "tests rc=$?" > $O/run14_status.txt
timeout 900 python scripts/wgrad_microbench.py 1 2 3 4 5 6 7 8 9 10 12 13 14 16 21 28 -1 0 > $O/run14_sweep.txt 2>&1
tail -3 $O/run14_tests.log; cat $O/run14_status.txt; cat $O/run14_sweep.txt
DFH_LIB=scripts/probes/build/libdifashion_probes.so timeout 600 python -m pytest scripts/probes/tests -x -q > $O/run14_probe_tests.log 2>&1; echo "probe tests rc=$?" >> $O/run14_status.txt
DFH_LIB=scripts/probes/build/libdifashion_probes.so timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k mlp_fused >> $O/run14_probe_tests.log 2>&1; echo "probe mlp rc=$?" >> $O/run14_status.txt
tail -3 $O/run14_probe_tests.log; cat $O/run14_status.txt
