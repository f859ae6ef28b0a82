#!/usr/bin/env python3
"""8x8-level 3x3 convs (M = 1024 at batch 16): tile / split-K choices after the eight-wave tile and the lean tap staging."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run
run("warm", 1024, 1280, 0, conv=(16, 8, 1280, 1, 0), resid=False); run("warm", 1024, 1280, 0, conv=(16, 8, 1280, 1, 0), resid=False)
run("conv 1280->1280 @8 [auto]", 1024, 1280, 0, conv=(16, 8, 1280, 1, 0), resid=False)
for tile, tag in ((1, "256x160r3"), (10, "128x160w8"), (4, "128x160s2")):
    for split in (2, 4, 8):
        run(f"conv 1280->1280 @8 [{tag} split {split}]", 1024, 1280, 0, conv=(16, 8, 1280, 1, 0), resid=False, tile=tile, split=split)
run("conv 2560->1280 @8 [auto]", 1024, 1280, 0, conv=(16, 8, 2560, 1, 0), resid=False)
for tile, tag in ((1, "256x160r3"), (10, "128x160w8")):
    for split in (4, 8):
        run(f"conv 2560->1280 @8 [{tag} split {split}]", 1024, 1280, 0, conv=(16, 8, 2560, 1, 0), resid=False, tile=tile, split=split)
run("conv 1280->1280 @16 [auto]", 4096, 1280, 0, conv=(16, 16, 1280, 1, 0), resid=False)
for split in (1, 2, 3, 4):
    run(f"conv 1280->1280 @16 [128x160w8 split {split}]", 4096, 1280, 0, conv=(16, 16, 1280, 1, 0), resid=False, tile=10, split=split)
