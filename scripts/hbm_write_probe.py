#!/usr/bin/env python3
"""Pure-write / copy / read rates of this box's HBM through plain torch kernels (context for the write-heavy GEMM launches)."""
import torch
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (64, 336, 2048):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device="cuda"); b = torch.empty_like(a)
    tw = t(lambda: a.fill_(1.0)); tc = t(lambda: b.copy_(a)); tr = t(lambda: a.sum())
    print(f"{mb:5d} MB: fill {mb / 1024 / tw:6.2f} GB/s ({tw * 1e6:7.1f} us)   copy (r+w) {2 * mb / 1024 / tc:6.2f} GB/s ({tc * 1e6:7.1f} us)   sum (read) {mb / 1024 / tr:6.2f} GB/s")
