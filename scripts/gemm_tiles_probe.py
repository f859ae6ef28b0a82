#!/usr/bin/env python3
"""Does throughput follow the staged bytes per flop?  One large plain GEMM under every tile variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run
for tile, tag, inten in ((3, "128x64 r3", 43), (5, "128x128 s2", 64), (4, "128x160 s2", 71), (2, "256x128 r3", 85), (1, "256x160 r3", 98)):
    run(f"plain 65536x1280x1280 [{tag}, {inten} flop/B]", 65536, 1280, 1280, tile=tile, bias=False, resid=False)
    run(f"plain 16384x2560x2560 [{tag}, {inten} flop/B]", 16384, 2560, 2560, tile=tile, bias=False, resid=False)
