#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.gemm_microbench import run
for tile, tag in ((4, "128x160s2"), (1, "256x160r3"), (5, "128x128s2"), (2, "256x128r3")):
    run(f"plain K2880 [{tag}]", 65536, 320 if tile in (4, 1) else 384, 2880, bias=False, resid=False, tile=tile)
    run(f"conv 320->320@64 [{tag}]", 65536, 320 if tile in (4, 1) else 384, 0, conv=(16, 64, 320, 1, 0), bias=False, resid=False, tile=tile)
run("plain K2880 N=1280 [128x160s2]", 16384, 1280, 2880, bias=False, resid=False, tile=4)
run("plain 8192x8192x4096 [128x160s2]", 8192, 8160, 4096, bias=False, resid=False, tile=4)
run("plain 8192x8192x4096 [256x160r3]", 8192, 8160, 4096, bias=False, resid=False, tile=1)
run("plain 8192x8192x4096 [128x128s2]", 8192, 8192, 4096, bias=False, resid=False, tile=5)
run("plain 8192x8192x4096 [256x128r3]", 8192, 8192, 4096, bias=False, resid=False, tile=2)
