#!/usr/bin/env python3
"""Where does the time of the short-K linears go?  K sweep (fixed cost vs per-k-step cost), epilogue variants, tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run

for K in (64, 128, 320, 640, 1280, 2560):
    run(f"K sweep N320 K={K} bias+resid", 65536, 320, K)
for K in (64, 320, 1280):
    run(f"K sweep N320 K={K} no epilogue", 65536, 320, K, bias=False, resid=False)
for tile, tag in ((3, "128x64r3"), (4, "128x160s2"), (5, "128x128s2"), (1, "256x160r3")):
    run(f"N320 K320 [{tag}]", 65536, 320, 320, tile=tile)
for N in (320, 640, 960, 2560):
    run(f"N sweep K320 N={N}", 65536, N, 320, bias=False, resid=False)
run("M=131072 N320 K320", 131072, 320, 320)
run("M=16384 N320 K320", 16384, 320, 320)
