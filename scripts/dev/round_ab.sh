# same-box A/B of the ROUND: the round-4 tree (gpurun_ab/r04: `git archive f3204e8`, built with its own Makefile; not in history, travels with gpurun) against HEAD, alternating
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  (cd gpurun_ab/r04 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('sampling  round-4 tree', d['ms_per_step'], d['value'])")
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('sampling  HEAD        ', d['ms_per_step'], d['value'])"
done
for i in 1 2; do
  (cd gpurun_ab/r04 && python bench.py --steps 20 --warmup 5 --dtype fp8 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fp8       round-4 tree', d['ms_per_step'], d['value'])")
  python bench.py --steps 20 --warmup 5 --dtype fp8 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fp8       HEAD        ', d['ms_per_step'], d['value'])"
done
for i in 1 2; do
  (cd gpurun_ab/r04 && python bench.py --mode train --steps 6 --warmup 2 --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('training  round-4 tree', d['ms_per_step'], d['value'])")
  python bench.py --mode train --steps 6 --warmup 2 --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('training  HEAD        ', d['ms_per_step'], d['value'])"
done
