#!/usr/bin/env python3
"""Cost of the transposed (V^T) epilogue vs the plain one on the attention V projections (batch 16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run
for (M, N, K, name) in ((65536, 320, 320, "64^2 C320"), (16384, 640, 640, "32^2 C640"), (4096, 1280, 1280, "16^2 C1280"), (1232, 12480, 768, "text V all layers")):
    run(f"V proj {name} plain", M, N, K, bias=False, resid=False)
    run(f"V proj {name} transposed", M, N, K, bias=False, resid=False, out_mode=1)
