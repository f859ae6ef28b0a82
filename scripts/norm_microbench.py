#!/usr/bin/env python3
"""GroupNorm(+SiLU) / LayerNorm kernels on the U-Net's shapes at batch 16: time and effective HBM GB/s (read + write once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difashion_amd import _lib

DEV = "cuda"
B = 16


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for H, C0, C1 in ((64, 320, 0), (64, 320, 320), (64, 640, 320), (32, 320, 0), (32, 640, 0), (32, 640, 320), (32, 640, 640), (32, 1280, 640), (16, 640, 0),
                  (16, 1280, 0), (16, 1280, 640), (16, 1280, 1280), (8, 1280, 0), (8, 1280, 1280)):
    HW = H * H
    C = C0 + C1
    x0 = torch.randn(B, HW, C0, device=DEV).bfloat16()
    x1 = torch.randn(B, HW, C1, device=DEV).bfloat16() if C1 else None
    g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    out = torch.empty(B, HW, C, device=DEV, dtype=torch.bfloat16)
    part = torch.empty(B * 64 * 64 * 2, device=DEV)
    s = _lib.stream_ptr()
    us = timeit(lambda: _lib.call("dfh_groupnorm", _lib.ptr(x0), C0, _lib.ptr(x1) if C1 else None, C1, B, HW, 32, _lib.ptr(g), _lib.ptr(b),
                                  1e-5, 1, _lib.ptr(out), _lib.ptr(part), s))
    by = 4.0 * B * HW * C
    print(f"groupnorm+silu {H}x{H} C={C0}+{C1}: {us:7.1f} us  {by / us / 1e3:7.1f} GB/s (r+w once)")
