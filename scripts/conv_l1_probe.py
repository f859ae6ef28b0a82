#!/usr/bin/env python3
"""32x32 / 16x16 / 8x8-level convs and deep linears: heuristic pick vs the 256 x 160 eight-wave ring (tile 1) and forced split factors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.gemm_microbench import run
shapes = [
    ("conv 640->640 @32", dict(M=16384, N=640, K=0, conv=(16, 32, 640, 1, 0), resid=False)),
    ("conv 1280->640 @32", dict(M=16384, N=640, K=0, conv=(16, 32, 1280, 1, 0), resid=False)),
    ("conv 320->640 @32", dict(M=16384, N=640, K=0, conv=(16, 32, 320, 1, 0), resid=False)),
    ("ffp 32^2 K3200 +res", dict(M=16384, N=640, K=3200)),
    ("conv 1280->1280 @16", dict(M=4096, N=1280, K=0, conv=(16, 16, 1280, 1, 0), resid=False)),
    ("conv 2560->1280 @16", dict(M=4096, N=1280, K=0, conv=(16, 16, 2560, 1, 0), resid=False)),
    ("conv 1280->1280 @8", dict(M=1024, N=1280, K=0, conv=(16, 8, 1280, 1, 0), resid=False)),
    ("conv 2560->1280 @8", dict(M=1024, N=1280, K=0, conv=(16, 8, 2560, 1, 0), resid=False)),
]
for rnd in range(2):
    for name, kw in shapes:
        for tile, split, tag in ((0, 0, "auto"), (1, 1, "256x160x8w s1"), (1, 2, "256x160x8w s2"), (1, 4, "256x160x8w s4"), (10, 1, "8w128 s1"), (10, 2, "8w128 s2"), (10, 4, "8w128 s4")):
            try:
                run(f"{name} [{tag}]", tile=tile, split=split, iters=20, warm=3, **kw)
            except Exception as e:
                print(name, tag, "ERR", str(e)[:80])
