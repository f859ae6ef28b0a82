#!/usr/bin/env python3
"""Linear class (the 16x16 / 32x32 / 64x64 transformer GEMMs): fixed cost vs per-k-step cost, tiles, split-K."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run

TILES = ((0, "auto"), (2, "t2"), (3, "128x64r3"), (4, "128x160s2"), (5, "128x128s2"), (6, "wide256x160"), (7, "wide128x160"), (9, "wide256x128"))
for K in (320, 640, 1280, 2560, 5120):
    run(f"16^2 N1280 K={K}", 4096, 1280, K)
for tile, tag in TILES:
    run(f"16^2 N1280 K1280 [{tag}]", 4096, 1280, 1280, tile=tile)
for split in (1, 2, 4):
    run(f"16^2 N1280 K1280 split={split}", 4096, 1280, 1280, split=split)
    run(f"16^2 N1280 K5120 split={split}", 4096, 1280, 5120, split=split)
for K in (320, 640, 1280, 2560):
    run(f"32^2 N640 K={K}", 16384, 640, K)
for tile, tag in TILES:
    run(f"32^2 N640 K640 [{tag}]", 16384, 640, 640, tile=tile)
    run(f"32^2 N1920 K640 qkv [{tag}]", 16384, 1920, 640, tile=tile, bias=False, resid=False)
for tile, tag in TILES:
    run(f"64^2 N320 K320 [{tag}]", 65536, 320, 320, tile=tile)
    run(f"64^2 N960 K320 qkv [{tag}]", 65536, 960, 320, tile=tile, bias=False, resid=False)
for tile, tag in TILES:
    run(f"geglu 16^2 N10240 K1280 [{tag}]", 4096, 10240, 1280, tile=tile, act=4)
    run(f"geglu 64^2 N2560 K320 [{tag}]", 65536, 2560, 320, tile=tile, act=4)
