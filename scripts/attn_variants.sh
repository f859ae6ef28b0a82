mkdir -p gpurun_out/r02
for v in 0 1 2; do echo "variant $v"; DFH_ATTN_VARIANT=$v python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from scripts.attn_microbench import run
run("self 64^2 d40", 16, 8, 40, 4096, 4096)
run("self 32^2 d80", 16, 8, 80, 1024, 1024)
PY
done
