#!/usr/bin/env python3
"""Per-shape timing of the fused attention kernel through the C ABI (diagnosis tool, GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difashion_amd import _lib
from scripts.gemm_microbench import timeit

DEV = "cuda"


def run(name, B, H, D, Nq, Nk):
    C_ = H * D
    q = torch.randn(B, Nq, C_, device=DEV).bfloat16()
    k = torch.randn(B, Nk, C_, device=DEV).bfloat16()
    ld = (Nk + 7) // 8 * 8
    vt = torch.randn(B, C_, ld, device=DEV).bfloat16()
    o = torch.empty_like(q)
    s = _lib.stream_ptr()
    us = timeit(lambda: _lib.call("dfh_attention", _lib.ptr(q), C_, _lib.ptr(k), C_, _lib.ptr(vt), ld, _lib.ptr(o), C_,
                                  B, H, D, Nq, Nk, D ** -0.5, s))
    fl = 4.0 * B * H * Nq * Nk * D
    print(f"{name:28s} B={B} H={H} D={D:3d} Nq={Nq:5d} Nk={Nk:5d} {us:9.1f} us {fl / us / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    run("self 64^2 d40", 16, 8, 40, 4096, 4096)
    run("self 32^2 d80", 16, 8, 80, 1024, 1024)
    run("self 16^2 d160", 16, 8, 160, 256, 256)
    run("self 8^2 d160", 16, 8, 160, 64, 64)
    run("cross 64^2 d40", 16, 8, 40, 4096, 77)
    run("cross 32^2 d80", 16, 8, 80, 1024, 77)
    run("cross 16^2 d160", 16, 8, 160, 256, 77)
    run("self 64^2 d64 (sd2)", 16, 5, 64, 4096, 4096)
