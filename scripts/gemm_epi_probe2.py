#!/usr/bin/env python3
"""Epilogue cost on the 32x32 / 16x16 levels (gemm.hip tiles): bias / residual on and off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_microbench import run

run("warm", 16384, 640, 640)
run("warm", 16384, 640, 640)
for bias, resid in ((False, False), (True, False), (True, True)):
    run(f"32^2 N640 K640 b={int(bias)} r={int(resid)}", 16384, 640, 640, bias=bias, resid=resid)
    run(f"32^2 N640 K2560 b={int(bias)} r={int(resid)}", 16384, 640, 2560, bias=bias, resid=resid)
    run(f"16^2 N1280 K1280 b={int(bias)} r={int(resid)}", 4096, 1280, 1280, bias=bias, resid=resid)
    run(f"16^2 N1280 K5120 b={int(bias)} r={int(resid)}", 4096, 1280, 5120, bias=bias, resid=resid)
    run(f"conv 640->640 @32 b={int(bias)} r={int(resid)}", 16384, 640, 0, conv=(16, 32, 640, 1, 0), bias=bias, resid=resid)
