#!/usr/bin/env python3
"""Epilogue cost of the chip-filling launches: the same GEMM with and without bias / residual, wide and 128-row tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run

for tile, tag in ((0, "auto"), (4, "128x160s2"), (6, "wide256x160")):
    for bias, resid in ((False, False), (True, False), (True, True)):
        run(f"64^2 N320 K320 b={int(bias)} r={int(resid)} [{tag}]", 65536, 320, 320, tile=tile, bias=bias, resid=resid)
    run(f"64^2 N320 K1280 ff2 [{tag}]", 65536, 320, 1280, tile=tile)
    run(f"conv 320->320 @64 +resid [{tag}]", 65536, 320, 0, conv=(16, 64, 320, 1, 0), tile=tile)
    run(f"32^2 N640 K640 [{tag}]", 16384, 640, 640, tile=tile)
