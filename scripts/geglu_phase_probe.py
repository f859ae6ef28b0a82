#!/usr/bin/env python3
"""Where a tile of the GEGLU projection spends its time: DFH_GEGLU_PROF=1 makes the persistent kernel (csrc/gemm_geglu.hip) stamp
s_memtime at its phase boundaries (workgroup 0, thread 0) and print per-phase cycle averages.  GPU only.
    DFH_GEGLU_ROWS=1 DFH_GEGLU_PROF=1 python scripts/geglu_phase_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.gemm_microbench import run

for name, M, C in (("ff1 64^2", 65536, 320), ("ff1 32^2", 16384, 640), ("ff1 16^2", 4096, 1280)):
    run(f"{name} auto", M, 8 * C, C, act=4, resid=False, tile=0)
