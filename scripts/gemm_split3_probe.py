#!/usr/bin/env python3
"""Split-K revisited after the epilogue fixes: auto pick vs forced split 1 / 2 / 4 on the 16x16 / 8x8-level launches (times include
the reduce kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.gemm_microbench import run

shapes = [
    ("conv 1280->1280 @16", dict(M=4096, N=1280, K=0, conv=(16, 16, 1280, 1, 0), resid=False)),
    ("conv 2560->1280 @16", dict(M=4096, N=1280, K=0, conv=(16, 16, 2560, 1, 0), resid=False)),
    ("conv 1920->1280 @16", dict(M=4096, N=1280, K=0, conv=(16, 16, 1920, 1, 0), resid=False)),
    ("lin 16^2 C1280 +res", dict(M=4096, N=1280, K=1280)),
    ("ff2 16^2 K5120 +res", dict(M=4096, N=1280, K=5120)),
    ("conv 1280->1280 @8", dict(M=1024, N=1280, K=0, conv=(16, 8, 1280, 1, 0), resid=False)),
    ("conv 2560->1280 @8", dict(M=1024, N=1280, K=0, conv=(16, 8, 2560, 1, 0), resid=False)),
    ("lin 8^2 C1280 +res", dict(M=1024, N=1280, K=1280)),
    ("ff2 8^2 K5120 +res", dict(M=1024, N=1280, K=5120)),
    ("conv 640->640 @32", dict(M=16384, N=640, K=0, conv=(16, 32, 640, 1, 0), resid=False)),
]
for name, kw in shapes:
    for split in (0, 1, 2, 4, 8):
        try:
            run(f"{name} [split {split or 'auto'}]", split=split, iters=30, warm=5, **kw)
        except Exception as e:
            print(name, split, "failed:", str(e)[:80])
