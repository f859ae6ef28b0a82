#!/usr/bin/env python3
"""One-tile-per-CU launches (16x16 level at batch 16): 2-stage ring vs the 4-stage ring (tile id 10)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run

run("warm", 4096, 1280, 1280); run("warm", 4096, 1280, 1280)
for tile, tag in ((4, "128x160s2"), (11, "128x160w8s3"), (12, "128x160w8s2"), (0, "auto")):
    for K in (320, 1280, 5120):
        run(f"16^2 N1280 K={K} [{tag}]", 4096, 1280, K, tile=tile)
    run(f"16^2 N1280 K1280 nores [{tag}]", 4096, 1280, 1280, tile=tile, bias=False, resid=False)
    run(f"8^2 N1280 K1280 [{tag}]", 1024, 1280, 1280, tile=tile)
    run(f"16^2 M8192 N1280 K1280 [{tag}]", 8192, 1280, 1280, tile=tile)
    run(f"32^2 N640 K640 [{tag}]", 16384, 640, 640, tile=tile)
    run(f"conv 1280->1280 @16 [{tag}]", 4096, 1280, 0, conv=(16, 16, 1280, 1, 0), tile=tile, resid=False)
    run(f"conv 1280->1280 @16 split1 [{tag}]", 4096, 1280, 0, conv=(16, 16, 1280, 1, 0), tile=tile, resid=False, split=1)
    run(f"conv 2560->1280 @16 split1 [{tag}]", 4096, 1280, 0, conv=(16, 16, 2560, 1, 0), tile=tile, resid=False, split=1)
