#!/usr/bin/env python3
"""SD-2-base attention shapes (head dim 64 at every level: the reference's own default model) at U-Net batch 16, through the C ABI.
    python scripts/attn_sd2_microbench.py            # whatever attention_launch picks
    DFH_ATTN_X32=0 python scripts/attn_sd2_microbench.py     # the 16x16x32 kernel (what ran before round 4)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.attn_microbench import run
run("sd2 self 64^2", 16, 5, 64, 4096, 4096)
run("sd2 self 32^2", 16, 10, 64, 1024, 1024)
run("sd2 self 16^2", 16, 20, 64, 256, 256)
run("sd2 self 8^2", 16, 20, 64, 64, 64)
run("sd2 cross 64^2", 16, 5, 64, 4096, 77)
run("sd2 cross 32^2", 16, 10, 64, 1024, 77)
run("sd2 cross 16^2", 16, 20, 64, 256, 77)
