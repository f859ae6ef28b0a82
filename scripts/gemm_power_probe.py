#!/usr/bin/env python3
"""Is the GEMM family power / clock limited?  Same launches on random and on zero-filled operands (MFMA power is data dependent)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run
for z in (False, True):
    tag = "zeros " if z else "random"
    run(f"conv 960->320 @64 wide [{tag}]", 65536, 320, 0, conv=(16, 64, 960, 1, 0), resid=False, tile=6, zeros=z)
    run(f"conv 960->320 @64 8-wave [{tag}]", 65536, 320, 0, conv=(16, 64, 960, 1, 0), resid=False, tile=8, zeros=z)
    run(f"conv 960->320 @64 s2 [{tag}]", 65536, 320, 0, conv=(16, 64, 960, 1, 0), resid=False, tile=4, zeros=z)
    run(f"plain 65536x1280x1280 wide [{tag}]", 65536, 1280, 1280, tile=6, bias=False, resid=False, zeros=z)
    run(f"plain 65536x1280x1280 8-wave [{tag}]", 65536, 1280, 1280, tile=8, bias=False, resid=False, zeros=z)
