#!/usr/bin/env python3
"""Why do the GEMMs run 30-40 % slower inside the step than back to back in a microbenchmark?  Same launch timed (HIP events around
the launch only) after (a) nothing, (b) a 1 GB streaming write that evicts L2 + Infinity Cache, (c) a different GEMM kernel
(instruction cache / clocks), (d) a GroupNorm-like elementwise kernel over the launch's own input (producer -> consumer warm input)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difashion_amd import _lib
DEV = "cuda"

def desc(M, N, K, resid=True, act=0):
    d = _lib.GemmDesc(); keep = []
    a = torch.randn(M, K, device=DEV).bfloat16(); w = (torch.randn(N, K, device=DEV) * 0.02).bfloat16(); b = torch.randn(N, device=DEV)
    d.a0, d.a0_c, d.W, d.ldw, d.M, d.N, d.bias = a.data_ptr(), K, w.data_ptr(), K, M, N, b.data_ptr()
    n_out = N // 2 if act == 4 else N
    if resid and act != 4:
        r = torch.randn(M, N, device=DEV).bfloat16(); keep.append(r); d.resid, d.ld_res = r.data_ptr(), N
    out = torch.empty(M, n_out, device=DEV, dtype=torch.bfloat16)
    d.out, d.ld_out, d.out_mode, d.act = out.data_ptr(), n_out, 0, act
    z = torch.zeros(256, dtype=torch.uint8, device=DEV); d.zero_page = z.data_ptr(); d.force_order = -1
    keep += [a, w, b, out, z]
    return d, keep, a

def timed(d, pre, iters=12):
    s = _lib.stream_ptr(); ts = []
    for i in range(iters + 3):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.call("dfh_gemm", C.byref(d), s); e1.record(); torch.cuda.synchronize()
        if i >= 3: ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]

junk = torch.empty(1 << 30, dtype=torch.uint8, device=DEV)
od, okeep, _ = desc(4096, 1280, 1280)
for name, (M, N, K, res, act) in {"lin 64^2 C320": (65536, 320, 320, False, 0), "lin 64^2 C320 +res": (65536, 320, 320, True, 0),
                                  "lin 32^2 C640 +res": (16384, 640, 640, True, 0), "ffp 32^2 K3200": (16384, 640, 3200, True, 0),
                                  "lin 16^2 C1280 +res": (4096, 1280, 1280, True, 0), "geglu 64^2": (65536, 2560, 320, False, 4),
                                  "ff2p 64^2 K1600": (65536, 320, 1600, True, 0)}.items():
    d, keep, a = desc(M, N, K, res, act)
    s = _lib.stream_ptr()
    r = dict(back_to_back=timed(d, lambda: None), after_1GB_write=timed(d, lambda: junk.fill_(1)),
             after_other_gemm=timed(d, lambda: _lib.call("dfh_gemm", C.byref(od), s)),
             after_touching_input=timed(d, lambda: a.mul_(1.0)))
    print(f"{name:22s} " + "  ".join(f"{k} {v:7.1f} us" for k, v in r.items()), flush=True)
