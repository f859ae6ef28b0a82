#!/usr/bin/env python3
"""Short-K linears (the launches whose prologue / epilogue outweigh the k-loop): default pick, the wide kernel and the eight-wave
128 x 160 kernel side by side."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.gemm_microbench import run

shapes = [
    ("lin 64^2 C320 +res", dict(M=65536, N=320, K=320)),
    ("lin 64^2 C320", dict(M=65536, N=320, K=320, resid=False)),
    ("ff2 64^2 K1280 +res", dict(M=65536, N=320, K=1280)),
    ("qkv 64^2 N960", dict(M=65536, N=960, K=320, bias=False, resid=False)),
    ("lin 32^2 C640 +res", dict(M=16384, N=640, K=640)),
    ("lin 32^2 C640", dict(M=16384, N=640, K=640, resid=False)),
    ("qkv 32^2 N1920", dict(M=16384, N=1920, K=640, bias=False, resid=False)),
    ("ff2 32^2 K2560 +res", dict(M=16384, N=640, K=2560)),
    ("lin 16^2 C1280 +res", dict(M=4096, N=1280, K=1280)),
    ("ff2 16^2 K5120 +res", dict(M=4096, N=1280, K=5120)),
    ("lin 8^2 C1280 +res", dict(M=1024, N=1280, K=1280)),
    ("conv 640->640 @32", dict(M=16384, N=640, K=0, conv=(16, 32, 640, 1, 0), resid=False)),
    ("conv 1280->1280 @16", dict(M=4096, N=1280, K=0, conv=(16, 16, 1280, 1, 0), resid=False)),
]
for name, kw in shapes:
    for tile, tag in ((0, "auto"), (6, "wide 256x160"), (10, "8-wave 128x160"), (4, "4-wave 128x160")):
        run(f"{name} [{tag}]", tile=tile, iters=30, warm=5, **kw)
