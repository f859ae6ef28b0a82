#!/usr/bin/env python3
"""Weight-gradient GEMM (gemm_wgrad_kernel) on the layer shapes of the SD-1.5 training step at batch 32.
   python scripts/wgrad_microbench.py [msplit ...]      (0 = the plan's choice, n = n pixel slices, -n = whole tiles in full rounds + the rest in n slices; several -> one table)"""
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
    ("conv 1920->1280 @16", 1, 16, 1920, 1280, 0),
    ("conv 1280->640 @32", 1, 32, 1280, 640, 0),
    ("conv 1920->640 @32", 1, 32, 1920, 640, 0),
    ("conv 960->640 @32", 1, 32, 960, 640, 0),
    ("conv 640->320 @64", 1, 64, 640, 320, 0),
    ("ff2 5120->1280 @16", 0, 16, 0, 1280, 5120),
    ("ff2 2560->640 @32", 0, 32, 0, 640, 2560),
    ("plain K=11520 N=1280 @16", 0, 16, 0, 1280, 11520),      # conv 1280->1280 @16 without the conv addressing
    ("plain K=2880 N=320 @64", 0, 64, 0, 320, 2880),
]


def main():
    plans = [int(x) for x in sys.argv[1:]] or [0]
    zero = torch.zeros(256, dtype=torch.uint8, device=DEV)
    s = _lib.stream_ptr()
    tot = {p: 0.0 for p in plans}
    tot_fl = 0.0
    print(f"{'shape':24s} {'M':>6s} {'N':>5s} {'K':>5s} {'tiles':>5s} | " + " ".join(f"{('ms=' + str(p)) if p > 0 else ('plan' if p == 0 else 'w+' + str(-p)):>8s}" for p in plans) + "   (us)")
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
        row = []
        for msplit in plans:
            if msplit > 1 and M // msplit < 256:
                row.append(float("nan")); continue
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
            row.append(ms * 1e3); tot[msplit] += ms
        fl = 2.0 * M * N * kk
        tot_fl += fl
        tiles = ((N + 159) // 160) * ((9 * ((cin + 159) // 160)) if conv else ((K + 159) // 160))
        best = min(r for r in row if r == r)
        print(f"{name:24s} {M:6d} {N:5d} {kk:5d} {tiles:5d} | " + " ".join(f"{r:8.1f}" for r in row) + f"   best {fl / best / 1e6:7.1f} TF", flush=True)
    print(f"{'sum (ms)':49s} | " + " ".join(f"{tot[p]:8.2f}" for p in plans))
    print(f"{'TFLOP/s':49s} | " + " ".join(f"{tot_fl / tot[p] / 1e9:8.1f}" for p in plans))


if __name__ == "__main__":
    main()
