import os, sys
sys.path.insert(0, "/root/repo")
from scripts.gemm_microbench import run
shapes = [
    ("conv 960->320 @64", dict(M=65536, N=320, K=0, conv=(16, 64, 960, 1, 0), resid=False)),
    ("conv 320->320 @64", dict(M=65536, N=320, K=0, conv=(16, 64, 320, 1, 0), resid=False)),
    ("ff2 64^2 K1280", dict(M=65536, N=320, K=1280, resid=False)),
    ("lin 64^2 K5120 N640", dict(M=65536, N=640, K=5120, resid=False)),
    ("lin 64^2 C320", dict(M=65536, N=320, K=320)),
    ("conv 640->640 @32", dict(M=16384, N=640, K=0, conv=(16, 32, 640, 1, 0), resid=False)),
]
for name, kw in shapes:
    for tile, tag in ((6, "wide 256x160 bk32"), (1, "bf16 256x160 bk64 8w 1wg"), (10, "bf16 128x160 bk64 8w 2wg"), (4, "bf16 128x160 4w")):
        for z in (False, True):
            run(f"{name} [{tag}{' zeros' if z else ''}]", tile=tile, iters=20, warm=3, zeros=z, **kw)
