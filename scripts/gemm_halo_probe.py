#!/usr/bin/env python3
"""The halo-patch 3x3 conv kernel (gemm_halo.hip, tile id 20) against the wide kernel (6) and the default pick, random and zero
operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.gemm_microbench import run

shapes = [
    ("conv 320->320 @64", dict(M=65536, N=320, K=0, conv=(16, 64, 320, 1, 0), resid=False)),
    ("conv 640->320 @64", dict(M=65536, N=320, K=0, conv=(16, 64, 640, 1, 0), resid=False)),
    ("conv 960->320 @64", dict(M=65536, N=320, K=0, conv=(16, 64, 960, 1, 0), resid=False)),
    ("conv 640->640 @32", dict(M=16384, N=640, K=0, conv=(16, 32, 640, 1, 0), resid=False)),
    ("conv 1920->640 @32", dict(M=16384, N=640, K=0, conv=(16, 32, 1920, 1, 0), resid=False)),
    ("conv 1280->1280 @16", dict(M=4096, N=1280, K=0, conv=(16, 16, 1280, 1, 0), resid=False)),
]
for name, kw in shapes:
    for tile, tag in ((0, "auto"), (6, "wide"), (20, "halo")):
        for z in (False, True):
            run(f"{name} [{tag}{' zeros' if z else ''}]", tile=tile, iters=20, warm=3, zeros=z, **kw)
