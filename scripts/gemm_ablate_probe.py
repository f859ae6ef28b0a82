#!/usr/bin/env python3
"""What paces the k-loop of the wide GEMM kernel?  The same launches with parts of the loop removed (tile ids 13..18, results are
garbage): 13 no LDS-DMA staging, 14 no fragment reads, 15 neither (MFMA only), 16 no MFMA, 17 fragment reads only, 18 staging only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.gemm_microbench import run

shapes = [
    ("conv 960->320 @64", dict(M=65536, N=320, K=0, conv=(16, 64, 960, 1, 0), resid=False)),
    ("conv 320->320 @64", dict(M=65536, N=320, K=0, conv=(16, 64, 320, 1, 0), resid=False)),
    ("ff2 64^2 K1280", dict(M=65536, N=320, K=1280, resid=False)),
    ("lin 64^2 K5120 N640", dict(M=65536, N=640, K=5120, resid=False)),
]
tags = {6: "full", 13: "no DMA", 14: "no LDS reads", 15: "MFMA only", 16: "no MFMA", 17: "LDS reads only", 18: "DMA only"}
for name, kw in shapes:
    for tile, tag in tags.items():
        run(f"{name} [{tag}]", tile=tile, iters=20, warm=3, **kw)
    for tile, tag in ((6, "full, zeros"), (15, "MFMA only, zeros")):
        run(f"{name} [{tag}]", tile=tile, iters=20, warm=3, zeros=True, **kw)
