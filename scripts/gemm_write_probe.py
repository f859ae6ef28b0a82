#!/usr/bin/env python3
"""Write-heavy linear launches (K = 320, M = 65536, N = 640 .. 2560): achieved bytes/s under column tiles of 160 (320-byte row pieces, every
second one splitting a 128-byte line) vs 128 / 256 columns.  python scripts/gemm_write_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run
for N in (640, 960, 1280, 2560):
    for t, tag in ((0, "auto"), (4, "128x160s2"), (5, "128x128s2"), (6, "256x160wide"), (21, "256x320big")):
        try:
            run(f"K=320 N={N} tile {tag}", 65536, N, 320, resid=False, tile=t)
        except Exception as e:
            print(f"K=320 N={N} tile {tag}: failed {str(e)[:80]}")
