#!/usr/bin/env python3
"""Split-K on the under-filled 16x16 / 8x8 levels (B=16): does doubling the grid pay for the slab round trip?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run
for split in (0, 2, 3, 4):
    run(f"conv 1280->1280 @16 split={split}", 4096, 1280, 0, conv=(16, 16, 1280, 1, 0), resid=False, split=split)
for split in (0, 2, 4):
    run(f"conv 2560->1280 @16 split={split}", 4096, 1280, 0, conv=(16, 16, 2560, 1, 0), resid=False, split=split)
for split in (0, 2, 4):
    run(f"linear 16^2 C1280 split={split}", 4096, 1280, 1280, split=split)
for split in (0, 2, 4, 8):
    run(f"conv 1280->1280 @8 split={split}", 1024, 1280, 0, conv=(16, 8, 1280, 1, 0), resid=False, split=split)
for split in (0, 2, 4, 8):
    run(f"linear 8^2 C1280 split={split}", 1024, 1280, 1280, split=split)
