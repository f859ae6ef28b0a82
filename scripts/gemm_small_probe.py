#!/usr/bin/env python3
"""Tile choice for the latency-bound linears of the 16x16 / 8x8 levels (batch 16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run
for tile, tag in ((0, "auto"), (3, "128x64r3"), (4, "128x160s2"), (5, "128x128s2")):
    run(f"lin 16^2 1280->1280 [{tag}]", 4096, 1280, 1280, tile=tile)
    run(f"qk 16^2 1280->2560 [{tag}]", 4096, 2560, 1280, tile=tile, bias=False, resid=False)
    run(f"ff2 16^2 5120->1280 [{tag}]", 4096, 1280, 5120, tile=tile)
    run(f"lin 8^2 1280->1280 [{tag}]", 1024, 1280, 1280, tile=tile)
    run(f"lin 32^2 640->640 [{tag}]", 16384, 640, 640, tile=tile)
run("ff1 16^2 geglu [auto]", 4096, 10240, 1280, act=4)
run("ff1 8^2 geglu [auto]", 1024, 10240, 1280, act=4)
