#!/usr/bin/env python3
"""The register-resident token-linear probe kernel (scripts/probes/kernels/token_linear.hip, tile id 30) against the tile GEMM on the 64x64-level
K = N = 320 projections of a sampling step (M = 65536).  GPU only; needs the probe library:
    make -C scripts/probes && DFH_LIB=scripts/probes/build/libdifashion_probes.so python scripts/token_linear_microbench.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from difashion_amd import _lib
import gpu_util as gu
from scripts.gemm_microbench import timeit
DEV = "cuda"
C = 320
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
x = torch.randn(M, C, device=DEV).bfloat16(); res = torch.randn(M, C, device=DEV).bfloat16()
w = (torch.randn(C, C, device=DEV) * 0.05).bfloat16(); bias = torch.randn(C, device=DEV)
sp = gu.stream()
for name, r in (("plain", None), ("+resid", res)):
    dt = gu.gemm_desc(M=M, N=C, W=w, ldw=C, a0=x, a0_c=C, bias=bias, resid=r, force_tile=30)
    dg = gu.gemm_desc(M=M, N=C, W=w, ldw=C, a0=x, a0_c=C, bias=bias, resid=r)
    tf = timeit(lambda: _lib.call("dfh_gemm", ctypes.byref(dt), sp))
    tg = timeit(lambda: _lib.call("dfh_gemm", ctypes.byref(dg), sp))
    print(f"M={M} {name:8s}: token_linear (incl. its image pack) {tf:7.1f} us   dfh_gemm {tg:7.1f} us", flush=True)
