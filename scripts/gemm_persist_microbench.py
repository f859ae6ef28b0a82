#!/usr/bin/env python3
"""Persistent 128 x 160 token-linear kernel (gemm_persist.hip, tile id 24) against the one-tile-per-workgroup LEAN kernel (tile id 10) and
the launcher's own choice (tile 0) on the short-K linear shapes of the SD-1.5 walk at batch 16.  python scripts/gemm_persist_microbench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run  # noqa: E402

SHAPES = [
    ("64x64 to_out   M=65536 N=320 K=320 +resid", dict(M=65536, N=320, K=320, resid=True)),
    ("64x64 to_q     M=65536 N=320 K=320", dict(M=65536, N=320, K=320, resid=False)),
    ("64x64 q|k|v    M=65536 N=960 K=320", dict(M=65536, N=960, K=320, resid=False, bias=False)),
    ("64x64 prefix   M=49152 N=320 K=320 +resid", dict(M=49152, N=320, K=320, resid=True)),
    ("32x32 to_out   M=16384 N=640 K=640 +resid", dict(M=16384, N=640, K=640, resid=True)),
    ("32x32 to_q     M=16384 N=640 K=640", dict(M=16384, N=640, K=640, resid=False)),
    ("32x32 q|k|v    M=16384 N=1920 K=640", dict(M=16384, N=1920, K=640, resid=False, bias=False)),
    ("32x32 ff2.pout M=16384 N=640 K=3200 +resid", dict(M=16384, N=640, K=3200, resid=True)),
    ("batch 64 to_out M=262144 N=320 K=320 +resid", dict(M=262144, N=320, K=320, resid=True)),
]
for name, kw in SHAPES:
    res = {}
    for tile, tag in ((10, "tile "), (24, "persist"), (0, "auto ")):
        res[tag] = run(f"{tag} {name}", tile=tile, iters=40, warm=5, **kw)["us"]
    print(f"    -> persistent / tile = {res['persist'] / res['tile ']:.3f}\n", flush=True)
