cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 300 python scripts/graph_probe.py > gpurun_out/r05/run28_graph.txt 2>&1
cat gpurun_out/r05/run28_graph.txt | tail -5
