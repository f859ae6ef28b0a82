#!/usr/bin/env python3
"""3x3 convs on the wide kernel (64x64-level shapes at batch 16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run
run("warm", 65536, 320, 0, conv=(16, 64, 320, 1, 0), resid=False); run("warm", 65536, 320, 0, conv=(16, 64, 320, 1, 0), resid=False)
run("conv 320->320 @64", 65536, 320, 0, conv=(16, 64, 320, 1, 0), resid=False)
run("conv 640->320 @64", 65536, 320, 0, conv=(16, 64, 640, 1, 0), resid=False)
run("conv 960->320 @64", 65536, 320, 0, conv=(16, 64, 960, 1, 0), resid=False)
run("conv 640->640 @32 [wide]", 16384, 640, 0, conv=(16, 32, 640, 1, 0), resid=False, tile=6)
run("conv 320->320 @64 +resid", 65536, 320, 0, conv=(16, 64, 320, 1, 0), resid=True)
run("conv 640->640 @32 [auto]", 16384, 640, 0, conv=(16, 32, 640, 1, 0), resid=False)
run("conv 1280->640 @32 [auto]", 16384, 640, 0, conv=(16, 32, 1280, 1, 0), resid=False)
run("conv 1280->1280 @16 [auto]", 4096, 1280, 0, conv=(16, 16, 1280, 1, 0), resid=False)
run("conv 2560->1280 @16 [auto]", 4096, 1280, 0, conv=(16, 16, 2560, 1, 0), resid=False)
run("conv 1280->1280 @8 [auto]", 1024, 1280, 0, conv=(16, 8, 1280, 1, 0), resid=False)
