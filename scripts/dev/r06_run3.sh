O=gpurun_out/r06_run3; mkdir -p $O
cat > /tmp/mb.py <<'PY'
import os, sys
sys.path.insert(0, "scripts")
from gemm_microbench import run
for name, kw in (("to_q 64x64", dict(M=65536, N=320, K=320, resid=False)), ("to_out 64x64 +resid", dict(M=65536, N=320, K=320, resid=True)), ("32x32 K3200", dict(M=16384, N=640, K=3200, resid=True))):
    run("persist " + name, tile=24, iters=40, warm=5, **kw)
PY
for d in 0 1 2 3 4 6 7; do echo "DFH_PERSIST_DBG=$d"; DFH_PERSIST_DBG=$d python /tmp/mb.py 2>&1 | grep -v amdgpu; done | tee $O/persist_dbg.txt
