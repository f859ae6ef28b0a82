#!/usr/bin/env python3
"""Weight-gradient GEMM (gemm_wgrad_kernel) on the layer shapes of the SD-1.5 training step at batch 32.
   python scripts/wgrad_microbench.py [msplit]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from difashion_amd import _lib

DEV = "cuda"
B = 32
SHAPES = [  # name, conv?, H, cin, cout (N), plain K
    ("conv 320->320 @64", 1, 64, 320, 320, 0),
    ("conv 640->640 @32", 1, 32, 640, 640, 0),
    ("conv 1280->1280 @16", 1, 16, 1280, 1280, 0),
    ("conv 1280->1280 @8", 1, 8, 1280, 1280, 0),
    ("conv 2560->1280 @8", 1, 8, 2560, 1280, 0),
    ("conv 960->320 @64", 1, 64, 960, 320, 0),
    ("lin 320->320 @64", 0, 64, 0, 320, 320),
    ("ff1 320->2560 @64", 0, 64, 0, 2560, 320),
    ("ff2 1280->320 @64", 0, 64, 0, 320, 1280),
    ("lin 640->640 @32", 0, 32, 0, 640, 640),
    ("ff1 640->5120 @32", 0, 32, 0, 5120, 640),
    ("lin 1280->1280 @16", 0, 16, 0, 1280, 1280),
    ("ff1 1280->10240 @16", 0, 16, 0, 10240, 1280),
    ("plain K=11520 N=1280 @16", 0, 16, 0, 1280, 11520),      # conv 1280->1280 @16 without the conv addressing
    ("plain K=2880 N=320 @64", 0, 64, 0, 320, 2880),
]


def main():
    msplit = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    zero = torch.zeros(256, dtype=torch.uint8, device=DEV)
    s = _lib.stream_ptr()
    tot_ms = tot_fl = 0.0
    for name, conv, H, cin, N, K in SHAPES:
        M = B * H * H
        d = _lib.GemmDesc()
        if conv:
            x = torch.randn(B, H, H, cin, device=DEV).to(torch.bfloat16)
            d.conv_src, d.conv_c, d.conv = x.data_ptr(), cin, 1
            d.batch, d.Hin, d.Win, d.stride, d.upsample = B, H, H, 1, 0
            kk = 9 * cin
        else:
            x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
            d.a0, d.a0_c = x.data_ptr(), K
            kk = K
        d.M, d.N, d.zero_page = M, N, zero.data_ptr()
        dy = torch.randn(M, N, device=DEV).to(torch.bfloat16)
        dw = torch.zeros(N, kk, device=DEV)
        need = _lib.raw().dfh_gemm_wgrad_partial_floats(C.byref(d), msplit)
        part = torch.empty(max(need, 1), dtype=torch.float32, device=DEV)
        d.partial, d.partial_floats = part.data_ptr(), need
        run = lambda: _lib.call("dfh_gemm_wgrad", C.byref(d), _lib.ptr(dy), N, _lib.ptr(dw), kk, msplit, s)
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 5
        e0.record()
        for _ in range(it):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / it
        fl = 2.0 * M * N * kk
        tot_ms += ms; tot_fl += fl
        print(f"{name:24s} M={M:6d} N={N:5d} K={kk:5d}  {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF")
    print(f"sum {tot_ms:.2f} ms  {tot_fl / tot_ms / 1e9:.1f} TF")


if __name__ == "__main__":
    main()
