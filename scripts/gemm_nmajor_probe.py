#!/usr/bin/env python3
"""m-major vs n-major tile ids (which operand an XCD's L2 shares) on the weight-heavy deep levels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run
for glds, tag in ((3, "m-major"), (2, "n-major")):
    run(f"conv 1280->1280 @16 B16 [{tag}]", 4096, 1280, 0, conv=(16, 16, 1280, 1, 0), resid=False, glds=glds)
    run(f"conv 2560->1280 @16 B16 [{tag}]", 4096, 1280, 0, conv=(16, 16, 2560, 1, 0), resid=False, glds=glds)
    run(f"conv 1280->1280 @8 B16 [{tag}]", 1024, 1280, 0, conv=(16, 8, 1280, 1, 0), resid=False, glds=glds)
    run(f"conv 1280->1280 @16 B32 [{tag}]", 8192, 1280, 0, conv=(32, 16, 1280, 1, 0), resid=False, glds=glds)
    run(f"linear 16^2 C1280 [{tag}]", 4096, 1280, 1280, glds=glds)
    run(f"ff2 16^2 5120->1280 [{tag}]", 4096, 1280, 5120, glds=glds)
    run(f"conv 640->640 @32 B16 [{tag}]", 16384, 640, 0, conv=(16, 32, 640, 1, 0), resid=False, glds=glds)
