#!/usr/bin/env python3
"""The VAE's conv shapes (N = 128 / 256 / 512) at batch 4: 128 x 128 tile of gemm.hip vs the 256 x 128 wide tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run
for tile, tag in ((5, "128x128s2"), (9, "256x128wide"), (0, "auto")):
    run(f"conv 128->128 @512 [{tag}]", 4 * 512 * 512, 128, 0, conv=(4, 512, 128, 1, 0), resid=False, tile=tile)
    run(f"conv 256->256 @256 [{tag}]", 4 * 256 * 256, 256, 0, conv=(4, 256, 256, 1, 0), resid=False, tile=tile)
    run(f"conv 512->512 @128 [{tag}]", 4 * 128 * 128, 512, 0, conv=(4, 128, 512, 1, 0), resid=False, tile=tile)
    run(f"conv 512->512 @64 [{tag}]", 4 * 64 * 64, 512, 0, conv=(4, 64, 512, 1, 0), resid=False, tile=tile)
