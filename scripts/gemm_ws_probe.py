#!/usr/bin/env python3
"""The wave-specialised GEMM (tile ids 11 / 12) against the wide kernel (6 / 9) and the default pick on the U-Net's big launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.gemm_microbench import run

shapes = [
    ("conv 320->320 @64", dict(M=65536, N=320, K=0, conv=(16, 64, 320, 1, 0), resid=False)),
    ("conv 960->320 @64", dict(M=65536, N=320, K=0, conv=(16, 64, 960, 1, 0), resid=False)),
    ("conv 640->640 @32", dict(M=16384, N=640, K=0, conv=(16, 32, 640, 1, 0), resid=False)),
    ("conv 1280->1280 @16", dict(M=4096, N=1280, K=0, conv=(16, 16, 1280, 1, 0), resid=False)),
    ("linear 64^2 C320", dict(M=65536, N=320, K=320)),
    ("linear 64^2 qk N640", dict(M=65536, N=640, K=320, resid=False, bias=False)),
    ("ff2 64^2 K1280", dict(M=65536, N=320, K=1280)),
    ("ff1 64^2 geglu", dict(M=65536, N=2560, K=320, act=4, resid=False)),
    ("ff1 32^2 geglu", dict(M=16384, N=5120, K=640, act=4, resid=False)),
    ("linear 32^2 C640", dict(M=16384, N=640, K=640)),
]
for name, kw in shapes:
    for tile in (6, 11, 0):
        if tile == 6 and kw.get("act") == 4:
            tile = 9
        run(f"{name} tile {tile}", tile=tile, **kw)
