#!/usr/bin/env python3
"""Lean plain k-loop (pointer-increment staging) vs the generic one (build with lean_plain() returning false) on the linear shapes of gemm.hip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run
run("warm", 4096, 1280, 1280); run("warm", 4096, 1280, 1280)
run("16^2 N1280 K1280", 4096, 1280, 1280)
run("16^2 N1280 K5120 ff2", 4096, 1280, 5120)
run("16^2 N2560 K1280 qk", 4096, 2560, 1280, bias=False, resid=False)
run("32^2 N640 K640", 16384, 640, 640)
run("32^2 N640 K2560 ff2", 16384, 640, 2560)
run("32^2 N1280 K640 qk", 16384, 1280, 640, bias=False, resid=False)
run("8^2 N1280 K1280", 1024, 1280, 1280)
run("64^2 N320 K320 [tile 10]", 65536, 320, 320, tile=10)
run("64^2 N320 K1280 [tile 10]", 65536, 320, 1280, tile=10)
