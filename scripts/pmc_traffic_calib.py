#!/usr/bin/env python3
"""PMC target: known-byte-count GEMM launches for calibrating FETCH_SIZE / WRITE_SIZE on this kernel family's access patterns
(LDS-DMA gathers of 128-byte and 64-byte rows) and for per-shape over-fetch.  Each shape is launched exactly N_IT times and
nothing else of the gemm family runs, so the aggregator (scripts/pmc_traffic_calib.sh) can cut the dispatch list into groups.
Writes the algorithmic bytes per shape to argv[1] (json) when given."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run

N_IT = 4
SHAPES = [
    # calibration: ONE column tile -> the activation matrix is read exactly once by construction, weights are 100 KB
    ("calib N160 128x160 (128-B rows)", dict(M=262144, N=160, K=320, bias=False, resid=False, tile=4)),
    ("calib N160 256x160 wide (64-B rows)", dict(M=262144, N=160, K=320, bias=False, resid=False, tile=6)),
    ("calib N128 256x128 wide (64-B rows)", dict(M=262144, N=128, K=320, bias=False, resid=False, tile=9)),
    ("ff1 64^2 geglu", dict(M=65536, N=2560, K=320, act=4, resid=False)),
    ("ff1 32^2 geglu", dict(M=16384, N=5120, K=640, act=4, resid=False)),
    ("ff1 16^2 geglu", dict(M=4096, N=10240, K=1280, act=4, resid=False)),
    ("linear 64^2 C320 +res", dict(M=65536, N=320, K=320)),
    ("ff2 64^2 K1280 +res", dict(M=65536, N=320, K=1280)),
    ("qkv 64^2 N960", dict(M=65536, N=960, K=320, bias=False, resid=False)),
    ("linear 32^2 C640 +res", dict(M=16384, N=640, K=640)),
    ("conv 320->320 @64", dict(M=65536, N=320, K=0, conv=(16, 64, 320, 1, 0), resid=False)),
    ("conv 960->320 @64", dict(M=65536, N=320, K=0, conv=(16, 64, 960, 1, 0), resid=False)),
    ("conv 640->640 @32", dict(M=16384, N=640, K=0, conv=(16, 32, 640, 1, 0), resid=False)),
    ("conv 1920->640 @32", dict(M=16384, N=640, K=0, conv=(16, 32, 1920, 1, 0), resid=False)),
    ("conv 1280->1280 @16", dict(M=4096, N=1280, K=0, conv=(16, 16, 1280, 1, 0), resid=False)),
    ("conv 2560->1280 @16", dict(M=4096, N=1280, K=0, conv=(16, 16, 2560, 1, 0), resid=False)),
    ("conv 1280->1280 @8", dict(M=1024, N=1280, K=0, conv=(16, 8, 1280, 1, 0), resid=False)),
]

if __name__ == "__main__":
    rec = []
    for name, kw in SHAPES:
        r = run(name, iters=N_IT, warm=0, **kw)
        rec.append(dict(name=name, launches=N_IT, **r))
    if len(sys.argv) > 1:
        json.dump(rec, open(sys.argv[1], "w"), indent=1)
