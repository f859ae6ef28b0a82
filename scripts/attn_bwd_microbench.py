#!/usr/bin/env python3
"""Attention backward (attention_bwd.hip: dQ pass + dK/dV pass) on the self-attention shapes of the SD-1.5 training step (batch 32 runs as
two launches of 16).  python scripts/attn_bwd_microbench.py        (DFH_LIB=<other .so> for a same-box A/B)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from difashion_amd import _lib
import gpu_util as gu
from scripts.gemm_microbench import timeit

DEV = "cuda"
for (B, H, d, N, Nk) in [(16, 8, 40, 4096, 4096), (16, 8, 80, 1024, 1024), (16, 8, 160, 256, 256), (16, 8, 40, 4096, 77), (16, 8, 64, 4096, 4096)]:
    C = H * d
    q, k, v, do = (torch.randn(B, n, C, device=DEV).bfloat16() for n in (N, Nk, Nk, N))
    ld = (Nk + 7) // 8 * 8
    vt = torch.zeros(B, C, ld, dtype=torch.bfloat16, device=DEV); vt[:, :, :Nk] = v.transpose(1, 2)
    o = torch.empty_like(q); lse = torch.empty(B, H, N, device=DEV); delta = torch.empty(B, H, N, device=DEV)
    sp = gu.stream()
    _lib.call("dfh_attention_lse", _lib.ptr(q), C, _lib.ptr(k), C, _lib.ptr(vt), ld, _lib.ptr(o), C, B, H, d, N, Nk, d ** -0.5, _lib.ptr(lse), sp)
    _lib.call("dfh_attention_delta", _lib.ptr(o), _lib.ptr(do), C, _lib.ptr(delta), B, H, d, N, sp)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    run = lambda: _lib.call("dfh_attention_bwd", _lib.ptr(q), C, _lib.ptr(k), C, _lib.ptr(v), C, _lib.ptr(do), C, _lib.ptr(lse), _lib.ptr(delta),
                            _lib.ptr(dq), C, _lib.ptr(dk), C, _lib.ptr(dv), C, B, H, d, N, Nk, d ** -0.5, sp)
    t = timeit(run)
    fl = 14.0 * B * H * N * Nk * d
    print(f"B={B} H={H} d={d:3d} Nq={N:5d} Nk={Nk:5d}: {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s (7 matmuls)", flush=True)
