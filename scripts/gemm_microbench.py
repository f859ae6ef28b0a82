#!/usr/bin/env python3
"""Per-shape timing of the implicit-GEMM kernel through the C ABI (diagnosis tool, GPU only)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difashion_amd import _lib

DEV = "cuda"


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
    return e0.elapsed_time(e1) / iters * 1e3   # us


def run(name, M, N, K, conv=None, bias=True, resid=True, out_mode=0, act=0, tile=0, split=0, glds=-1, zeros=False, iters=20, warm=3):
    d = _lib.GemmDesc()
    keep = []
    if conv:
        B, H, cin, stride, ups = conv
        x = torch.randn(B, H, H, cin, device=DEV).bfloat16(); keep.append(x)
        if zeros: x.zero_()
        d.conv_src, d.conv_c, d.conv, d.batch, d.Hin, d.Win, d.stride, d.upsample = x.data_ptr(), cin, 1, B, H, H, stride, ups
        Kt = 9 * cin
    else:
        a = torch.randn(M, K, device=DEV).bfloat16(); keep.append(a)
        if zeros: a.zero_()
        d.a0, d.a0_c = a.data_ptr(), K
        Kt = K
    w = (torch.randn(N, Kt, device=DEV) * 0.05).bfloat16(); keep.append(w)
    if zeros: w.zero_()
    d.W, d.ldw, d.M, d.N = w.data_ptr(), Kt, M, N
    if bias:
        b = torch.randn(N, device=DEV); keep.append(b); d.bias = b.data_ptr()
    n_out = N // 2 if act == 4 else N
    if resid and act != 4:
        r = torch.randn(M, N, device=DEV).bfloat16(); keep.append(r); d.resid, d.ld_res = r.data_ptr(), N
    out = torch.empty(M, n_out, device=DEV, dtype=torch.float32 if out_mode == 2 else torch.bfloat16); keep.append(out)
    d.out, d.ld_out, d.out_mode, d.act = out.data_ptr(), n_out, out_mode, act
    if out_mode in (1, 3):            # transposed per batch: [b][n][rows_per_b], 16 batches
        d.rows_per_b, d.ld_out = M // 16, M // 16
    z = torch.zeros(256, dtype=torch.uint8, device=DEV); keep.append(z); d.zero_page = z.data_ptr()
    d.force_tile, d.force_split, d.force_order = tile, split, glds
    need = _lib.raw().dfh_gemm_partial_floats(C.byref(d))
    if need:
        p = torch.empty(need, device=DEV); keep.append(p); d.partial, d.partial_floats = p.data_ptr(), need
    s = _lib.stream_ptr()
    us = timeit(lambda: _lib.call("dfh_gemm", C.byref(d), s), iters=iters, warm=warm)
    fl = 2.0 * M * N * Kt
    by = 2.0 * (M * Kt if not conv else conv[0] * conv[1] ** 2 * conv[2]) + 2.0 * N * Kt + out.numel() * out.element_size()
    if resid and act != 4:
        by += 2.0 * M * N
    print(f"{name:42s} M={M:6d} N={N:5d} K={Kt:6d} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s  {by / us / 1e3:7.1f} GB/s", flush=True)
    return dict(us=us, read_bytes=by - out.numel() * out.element_size(), write_bytes=out.numel() * out.element_size())


if __name__ == "__main__":
    print(_lib.raw().dfh_build_info().decode())
    shapes = [
        ("linear 64^2 C320", dict(M=65536, N=320, K=320)),
        ("linear 64^2 C320 K1280 (ff2)", dict(M=65536, N=320, K=1280)),
        ("linear 64^2 qk N640", dict(M=65536, N=640, K=320, resid=False, bias=False)),
        ("linear 32^2 C640", dict(M=16384, N=640, K=640)),
        ("linear 16^2 C1280", dict(M=4096, N=1280, K=1280)),
        ("linear 8^2 C1280", dict(M=1024, N=1280, K=1280)),
        ("cross kv 1232x320x768", dict(M=1232, N=320, K=768, bias=False, resid=False)),
        ("conv 320->320 @64", dict(M=65536, N=320, K=0, conv=(16, 64, 320, 1, 0), resid=False)),
        ("conv 960->320 @64", dict(M=65536, N=320, K=0, conv=(16, 64, 960, 1, 0), resid=False)),
        ("conv 640->640 @32", dict(M=16384, N=640, K=0, conv=(16, 32, 640, 1, 0), resid=False)),
        ("conv 1280->1280 @16", dict(M=4096, N=1280, K=0, conv=(16, 16, 1280, 1, 0), resid=False)),
        ("conv 1280->1280 @8", dict(M=1024, N=1280, K=0, conv=(16, 8, 1280, 1, 0), resid=False)),
    ]
    for name, kw in shapes:
        for tile, tag in ((0, "auto"), (4, "128x160s2"), (6, "256x160wide"), (8, "256x320x8w")):
            run(f"{name} [{tag}]", tile=tile, **kw)
    run("geglu 64^2 N2560 [auto]", 65536, 2560, 320, act=4)
    run("geglu 64^2 N2560 [128x128s2]", 65536, 2560, 320, act=4, tile=5)
    run("geglu 64^2 N2560 [wide]", 65536, 2560, 320, act=4, tile=6)
    run("geglu 32^2 N5120 [128x128s2]", 16384, 5120, 640, act=4, tile=5)
    run("geglu 32^2 N5120 [auto]", 16384, 5120, 640, act=4)
    run("temb 16x20480x1280", 16, 20480, 1280, resid=False, out_mode=2)
    run("conv 8->320 @64 (conv_in)", 65536, 320, 0, conv=(16, 64, 8, 1, 0), resid=False)
    run("conv 320->4 @64 (conv_out)", 65536, 4, 0, conv=(16, 64, 320, 1, 0), resid=False)
