# same-box A/B of the training step between two TREES (ABI changed: the previous build cannot be loaded through DFH_LIB): gpurun_ab/head = git archive HEAD
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  (cd gpurun_ab/head && python bench.py --mode train --steps 6 --warmup 2 --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('prev', d['ms_per_step'], d['loss'])")
  python bench.py --mode train --steps 6 --warmup 2 --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new ', d['ms_per_step'], d['loss'])"
done
