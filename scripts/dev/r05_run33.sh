cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
rm -f $O/run33_ab.txt
for i in 1 2 3; do
DFH_GN_FOLD=640 timeout 300 python bench.py --no-secondary --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fold 640', d['ms_per_step'])" >> $O/run33_ab.txt
timeout 300 python bench.py --no-secondary --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fold 320', d['ms_per_step'])" >> $O/run33_ab.txt
done
cat $O/run33_ab.txt
