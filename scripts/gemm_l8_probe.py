#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run
for glds, tag in ((3, "m-major"), (2, "n-major")):
    for split in (2, 4, 6, 8):
        run(f"conv 1280->1280 @8 B16 [{tag} split {split}]", 1024, 1280, 0, conv=(16, 8, 1280, 1, 0), resid=False, glds=glds, split=split)
run("conv 1280->1280 @8 B16 [auto]", 1024, 1280, 0, conv=(16, 8, 1280, 1, 0), resid=False)
for glds, tag in ((3, "m-major"), (2, "n-major")):
    for split in (2, 4):
        run(f"conv 1280->1280 @8 B32 [{tag} split {split}]", 2048, 1280, 0, conv=(32, 8, 1280, 1, 0), resid=False, glds=glds, split=split)
run("conv 1280->1280 @8 B32 [auto]", 2048, 1280, 0, conv=(32, 8, 1280, 1, 0), resid=False)
