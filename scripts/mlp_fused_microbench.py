#!/usr/bin/env python3
"""The fused GEGLU feed-forward + proj_out kernel (csrc/mlp_fused2.hip; FORM=1: the probe kernel, with the probe library) against the two launches it replaces, at the 64x64-level shape of a
sampling step (M = 16 x 4096 tokens, C = 320), through the C ABI.  GPU only.
    python scripts/mlp_fused_microbench.py [M]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from difashion_amd import _lib
import gpu_util as gu
from scripts.gemm_microbench import timeit

DEV = "cuda"
FORM = int(os.environ.get("FORM", 2))
C = 320
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
x = (torch.randn(M, C, device=DEV)).bfloat16()
resid = torch.randn(M, C, device=DEV).bfloat16()
wf = (torch.randn(8 * C, C, device=DEV) * 0.05).bfloat16()
s1, b1 = wf.float().sum(1).contiguous(), torch.randn(8 * C, device=DEV) * 0.1
w2p = (torch.randn(C, 5 * C, device=DEV) * 0.03).bfloat16()
bias = torch.randn(C, device=DEV)
parts, cnt = 2, 160
xp = x.float().view(M, parts, cnt).transpose(0, 1)
mean_t = xp.mean(-1)
st = torch.stack([mean_t, ((xp - mean_t[..., None]) ** 2).sum(-1)], dim=-1).contiguous()
img = torch.empty(_lib.raw().dfh_mlp_fused_image_bytes(), dtype=torch.uint8, device=DEV)
_lib.call("dfh_mlp_fused_pack", _lib.ptr(wf), _lib.ptr(s1), _lib.ptr(b1), _lib.ptr(w2p), _lib.ptr(img), FORM, gu.stream())
out = torch.empty((M, C), dtype=torch.bfloat16, device=DEV)
sp = gu.stream()
fused = lambda: _lib.call("dfh_mlp_fused", _lib.ptr(x), _lib.ptr(resid), _lib.ptr(img), _lib.ptr(st), parts, cnt, 1e-5, _lib.ptr(bias), _lib.ptr(out), M, FORM, None, 0, 0, sp)
d1 = gu.gemm_desc(M=M, N=8 * C, W=wf, ldw=C, a0=x, a0_c=C, bias=b1, act=4)
d2 = gu.gemm_desc(M=M, N=C, W=w2p, ldw=5 * C, a0=d1.keep_out, a0_c=4 * C, a1=x, a1_c=C, bias=bias, resid=resid)
def two():
    _lib.call("dfh_gemm_ln", ctypes.byref(d1), None, None, _lib.ptr(st), parts, cnt, 1e-5, _lib.ptr(s1), sp)
    _lib.call("dfh_gemm", ctypes.byref(d2), sp)
fl = 2.0 * M * (8 * C * C + 5 * C * C)
t_f, t_2 = timeit(fused), timeit(two)
torch.cuda.synchronize()
err = float((out.float() - d2.keep_out.float()).norm() / d2.keep_out.float().norm())
print(f"M={M} C={C}: fused {t_f:8.1f} us ({fl / t_f / 1e6:7.1f} TFLOP/s)   two launches {t_2:8.1f} us ({fl / t_2 / 1e6:7.1f} TFLOP/s)   rel diff {err:.2e}", flush=True)
