#!/usr/bin/env python3
"""Tile-order probe: the same launches under DFH_TMAP="xm,gm" settings (run once per setting: the env var is read once).
Usage: DFH_TMAP=2,8 python scripts/tile_order_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.gemm_microbench import run

print("DFH_TMAP =", os.environ.get("DFH_TMAP", "(legacy)"))
shapes = [
    ("ff1 64^2 geglu", dict(M=65536, N=2560, K=320, act=4, resid=False)),
    ("ff1 32^2 geglu", dict(M=16384, N=5120, K=640, act=4, resid=False)),
    ("ff1 16^2 geglu", dict(M=4096, N=10240, K=1280, act=4, resid=False)),
    ("qkv 64^2 N960", dict(M=65536, N=960, K=320, bias=False, resid=False)),
    ("qkv 32^2 N1920", dict(M=16384, N=1920, K=640, bias=False, resid=False)),
    ("linear 64^2 C320 +res", dict(M=65536, N=320, K=320)),
    ("ff2 64^2 K1280 +res", dict(M=65536, N=320, K=1280)),
    ("ff2 32^2 K2560 +res", dict(M=16384, N=640, K=2560)),
    ("conv 320->320 @64", dict(M=65536, N=320, K=0, conv=(16, 64, 320, 1, 0), resid=False)),
    ("conv 960->320 @64", dict(M=65536, N=320, K=0, conv=(16, 64, 960, 1, 0), resid=False)),
    ("conv 640->640 @32", dict(M=16384, N=640, K=0, conv=(16, 32, 640, 1, 0), resid=False)),
    ("conv 1920->640 @32", dict(M=16384, N=640, K=0, conv=(16, 32, 1920, 1, 0), resid=False)),
    ("conv 1280->1280 @16", dict(M=4096, N=1280, K=0, conv=(16, 16, 1280, 1, 0), resid=False)),
    ("conv 2560->1280 @16", dict(M=4096, N=1280, K=0, conv=(16, 16, 2560, 1, 0), resid=False)),
]
for name, kw in shapes:
    run(name, iters=30, warm=5, **kw)
