#!/usr/bin/env python3
"""FF1 (GEGLU) launches per level: default pick vs forced tiles (6 = 256x160 wide with the fp32 LDS epilogue, 9 = 256x128 wide with
the in-register GEGLU epilogue, 5 = 128x128 four-wave).  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.gemm_microbench import run

for name, M, C in (("ff1 64^2", 65536, 320), ("ff1 32^2", 16384, 640), ("ff1 16^2", 4096, 1280), ("ff1 8^2", 1024, 1280)):
    for tile in (6, 9, 5, 0, 9, 0):
        if tile == 6 and (8 * C) % 160:
            continue
        run(f"{name} tile {tile}", M, 8 * C, C, act=4, resid=False, tile=tile)
