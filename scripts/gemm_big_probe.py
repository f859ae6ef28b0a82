#!/usr/bin/env python3
"""The 256 x 320 eight-wave tile (tile id 21) against the heuristic pick on the launches that give every CU one tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.gemm_microbench import run

shapes = [
    ("conv 320->320 @64", dict(M=65536, N=320, K=0, conv=(16, 64, 320, 1, 0), resid=False)),
    ("conv 320->320 @64 +res", dict(M=65536, N=320, K=0, conv=(16, 64, 320, 1, 0), resid=True)),
    ("conv 640->320 @64", dict(M=65536, N=320, K=0, conv=(16, 64, 640, 1, 0), resid=False)),
    ("conv 960->320 @64", dict(M=65536, N=320, K=0, conv=(16, 64, 960, 1, 0), resid=False)),
    ("conv 8->320 @64 (conv_in)", dict(M=65536, N=320, K=0, conv=(16, 64, 8, 1, 0), resid=False)),
    ("ff2 64^2 K1280 +res", dict(M=65536, N=320, K=1280)),
    ("lin 64^2 C320 +res", dict(M=65536, N=320, K=320)),
    ("lin 64^2 C320", dict(M=65536, N=320, K=320, resid=False)),
    ("qk 64^2 N640", dict(M=65536, N=640, K=320, bias=False, resid=False)),
    ("conv 640->640 @32", dict(M=16384, N=640, K=0, conv=(16, 32, 640, 1, 0), resid=False)),
    ("conv 1280->640 @32", dict(M=16384, N=640, K=0, conv=(16, 32, 1280, 1, 0), resid=False)),
]
shapes += [
    ("conv 640->640 @32 +res", dict(M=16384, N=640, K=0, conv=(16, 32, 640, 1, 0), resid=True)),
    ("conv 320->640 @32", dict(M=16384, N=640, K=0, conv=(16, 32, 320, 1, 0), resid=False)),
    ("conv 1920->640 @32", dict(M=16384, N=640, K=0, conv=(16, 32, 1920, 1, 0), resid=False)),
    ("ff2 32^2 K2560 +res", dict(M=16384, N=640, K=2560)),
    ("lin 32^2 C640", dict(M=16384, N=640, K=640, resid=False)),
    ("lin 32^2 C640 +res", dict(M=16384, N=640, K=640)),
    ("qk 32^2 N1280", dict(M=16384, N=1280, K=640, bias=False, resid=False)),
]
from scripts.gemm_microbench import run as _run
for rnd in range(2):       # the GEGLU projections: heuristic pick, the 256 x 256 eight-wave tile (23), the wide 256 x 128 tile (9)
    for name, kw in (("geglu 64^2 N2560", dict(M=65536, N=2560, K=320)), ("geglu 32^2 N5120", dict(M=16384, N=5120, K=640)),
                     ("geglu 16^2 N10240", dict(M=4096, N=10240, K=1280))):
        for tile, tag in ((0, "auto"), (23, "256x256"), (9, "wide 256x128")):
            _run(f"{name} [{tag}]", tile=tile, act=4, iters=30, warm=5, **kw)
for rnd in range(0):
    for name, kw in shapes:
        for tile, tag in ((0, "auto"), (21, "256x320"), (6, "wide 256x160"), (10, "8-wave 128x160")):
            run(f"{name} [{tag}]", tile=tile, iters=30, warm=5, **kw)
