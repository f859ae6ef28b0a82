#!/usr/bin/env python3
"""One GEMM shape a few times (PMC target): python scripts/gemm_one.py conv320|lin320|conv1280|lin320k320|lin960k320|lin320k320_tile|lin320k320_persist"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run
which = sys.argv[1] if len(sys.argv) > 1 else "conv320"
if which == "conv320":
    run("conv 320->320 @64", 65536, 320, 0, conv=(16, 64, 320, 1, 0), resid=False)
elif which == "conv1280":
    run("conv 1280->1280 @16", 4096, 1280, 0, conv=(16, 16, 1280, 1, 0), resid=False)
elif which == "lin320k320":
    run("linear 64^2 N=K=320 plain", 65536, 320, 320, resid=False)
elif which == "lin320k320_tile":          # round 6: the one-tile-per-workgroup LEAN kernel pinned (tile id 10) against ...
    run("linear 64^2 N=K=320 plain, tile kernel", 65536, 320, 320, resid=False, tile=10)
elif which == "lin320k320_persist":       # ... the persistent probe kernel (tile id 24; needs DFH_LIB=<probe library>)
    run("linear 64^2 N=K=320 plain, persistent", 65536, 320, 320, resid=False, tile=24)
elif which == "lin960k320":
    run("linear 64^2 N=960 K=320 plain", 65536, 960, 320, resid=False, bias=False)
else:
    run("linear 64^2 C320 K1280", 65536, 320, 1280)
