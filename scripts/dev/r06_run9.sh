# round 6, GPU run 9: software-pipelined persistent kernel -- parity, microbench, timing probes, in-step A/B (probe library: DFH_PERSIST=1)
O=gpurun_out/r06_run9; mkdir -p $O
export DFH_LIB=$GRAFT_REPO_ROOT/scripts/probes/build/libdifashion_probes.so
timeout 600 python -m pytest scripts/probes/tests -q -x -k "persistent" 2>&1 | tail -4
timeout 600 python scripts/gemm_persist_microbench.py 2>&1 | grep -v amdgpu > $O/persist_microbench.txt; grep "persist\|->" $O/persist_microbench.txt
for d in 7 6 3; do echo "DFH_PERSIST_DBG=$d"; DFH_PERSIST_DBG=$d python /dev/stdin <<'PY' 2>&1 | grep -v amdgpu
import sys
sys.path.insert(0, "scripts")
from gemm_microbench import run
for name, kw in (("to_q 64x64", dict(M=65536, N=320, K=320, resid=False)), ("32x32 K3200", dict(M=16384, N=640, K=3200, resid=True))):
    run("persist " + name, tile=24, iters=40, warm=5, **kw)
PY
done | tee $O/persist_dbg.txt
for i in 1 2; do
  for m in 0 1; do
    DFH_PERSIST=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('probe lib DFH_PERSIST=$m', d['ms_per_step'])"
  done
done | tee $O/instep_ab.txt
