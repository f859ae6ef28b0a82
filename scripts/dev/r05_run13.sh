cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -x -q > $O/run13_tests.log 2>&1; echo "tests rc=$?" > $O/run13_status.txt
DFH_LIB=scripts/probes/build/libdifashion_probes.so timeout 600 python -m pytest scripts/probes/tests -x -q > $O/run13_probe_tests.log 2>&1; echo "probe tests rc=$?" >> $O/run13_status.txt
DFH_LIB=scripts/probes/build/libdifashion_probes.so timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k mlp_fused >> $O/run13_probe_tests.log 2>&1; echo "probe mlp rc=$?" >> $O/run13_status.txt
timeout 600 python bench.py --no-secondary > $O/run13_bench.json 2> $O/run13_bench.err
tail -3 $O/run13_tests.log; tail -3 $O/run13_probe_tests.log; cat $O/run13_status.txt; cat $O/run13_bench.json | cut -c1-400
