#!/usr/bin/env python3
"""Split-K choices on the 8x8 level at the training batch (32 items: M = 2048) and the 16x16 level (M = 8192)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run
for split in (0, 2, 3, 4, 6):
    run(f"conv 1280->1280 @8 B32 split={split}", 2048, 1280, 0, conv=(32, 8, 1280, 1, 0), resid=False, split=split)
for split in (0, 2, 4):
    run(f"conv 2560->1280 @8 B32 split={split}", 2048, 1280, 0, conv=(32, 8, 2560, 1, 0), resid=False, split=split)
for split in (0, 2):
    run(f"conv 1280->1280 @16 B32 split={split}", 8192, 1280, 0, conv=(32, 16, 1280, 1, 0), resid=False, split=split)
for split in (0, 2, 4):
    run(f"linear 8^2 C1280 B32 split={split}", 2048, 1280, 1280, split=split)
