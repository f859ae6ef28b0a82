#!/usr/bin/env python3
"""How the short-K linear launches (LEAN 128 x 160 tile) scale with rows and depth: time vs M at K = N = 320, time vs K at M = 65536.
Separates a fixed per-launch cost from a per-tile cost.  python scripts/gemm_scaling_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run
for M in (2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144):
    run(f"N=K=320 plain M={M}", M, 320, 320, resid=False)
for M in (32768, 65536, 131072):
    run(f"N=K=320 +resid M={M}", M, 320, 320, resid=True)
for K in (64, 128, 192, 320, 640, 1280, 2560):
    run(f"N=320 M=65536 K={K}", 65536, 320, K, resid=False)
for N in (160, 320, 640, 960, 1280):
    run(f"K=320 M=65536 N={N}", 65536, N, 320, resid=False)
for t, tag in ((0, "auto"), (4, "128x160s2"), (6, "256x160wide"), (8, "256x320x8w"), (21, "256x320big")):
    try:
        run(f"N=K=320 M=65536 tile {tag}", 65536, 320, 320, resid=False, tile=t)
    except Exception as e:
        print("tile", tag, "failed:", str(e)[:100])
