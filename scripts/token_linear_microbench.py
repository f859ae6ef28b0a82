#!/usr/bin/env python3
"""dfh_token_linear against dfh_gemm on the 64x64-level K = N = 320 projections of a sampling step (M = 65536).  GPU only."""
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
img = torch.empty(_lib.raw().dfh_token_linear_image_bytes(), dtype=torch.uint8, device=DEV)
_lib.call("dfh_token_linear_pack", _lib.ptr(w), C, _lib.ptr(img), gu.stream())
out = torch.empty((M, C), dtype=torch.bfloat16, device=DEV); rs = torch.empty((M, 2), device=DEV)
sp = gu.stream()
for name, r, st in (("plain", None, None), ("+resid", res, None), ("+resid +rowstat", res, rs)):
    f = lambda: _lib.call("dfh_token_linear", _lib.ptr(x), _lib.ptr(img), _lib.ptr(bias), _lib.ptr(r), None, 0, 0, 0.0, None, _lib.ptr(st), _lib.ptr(out), M, sp)
    d = gu.gemm_desc(M=M, N=C, W=w, ldw=C, a0=x, a0_c=C, bias=bias, resid=r)
    g = lambda: _lib.call("dfh_gemm", ctypes.byref(d), sp)
    tf, tg = timeit(f), timeit(g)
    print(f"M={M} {name:16s}: token_linear {tf:7.1f} us   dfh_gemm {tg:7.1f} us", flush=True)
