#!/bin/bash
# attention_x32_kernel variants (DFH_ATTN_VARIANT: 0 = default <40,2,2>, 1 / 2 = one query block per wave at 3 / 4 waves per SIMD,
# 3 = four query blocks per wave at one wave per SIMD) on the self- and cross-attention launches of the 64x64 / 32x32 levels.
mkdir -p gpurun_out/r02
for v in 0 1 2 3; do echo "variant $v"; DFH_ATTN_VARIANT=$v python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from scripts.attn_microbench import run
run("self 64^2 d40", 16, 8, 40, 4096, 4096)
run("cross 64^2 d40", 16, 8, 40, 4096, 77)
run("self 32^2 d80", 16, 8, 80, 1024, 1024)
run("cross 32^2 d80", 16, 8, 80, 1024, 77)
PY
done
