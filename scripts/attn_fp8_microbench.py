#!/usr/bin/env python3
"""bf16 attention kernels against attention_fp8_kernel on the self-attention shapes of the SD-1.5 U-Net at batch 16 (interleaved rounds,
HIP events, random operands; operand factors from random LayerNorm-folded weights as in tests/test_gpu_ops.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from difashion_amd import _lib
from tests import gpu_util as gu
from tests.gpu_util import DEV, bf, rnd
B = 16
for H, D, N in ((8, 40, 4096), (8, 80, 1024), (8, 160, 256), (8, 160, 64)):
    Cc = H * D
    x = torch.randn(B * N, Cc, device=DEV)
    gamma, beta = torch.ones(Cc, device=DEV), torch.zeros(Cc, device=DEV)
    w = bf(rnd(3 * Cc, Cc, seed=1, scale=0.05))
    wf = torch.empty_like(w); sv = torch.empty(3 * Cc, device=DEV); bv = torch.empty(3 * Cc, device=DEV)
    _lib.call("dfh_ln_fold", _lib.ptr(w), Cc, _lib.ptr(gamma), _lib.ptr(beta), None, _lib.ptr(wf), _lib.ptr(sv), _lib.ptr(bv), 3 * Cc, Cc, gu.stream())
    rq, rk, rv, hs = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV), torch.empty(H, device=DEV)
    _lib.call("dfh_attn_scales", _lib.ptr(wf), _lib.ptr(bv), Cc, H, _lib.ptr(rq), _lib.ptr(rk), _lib.ptr(rv), _lib.ptr(hs), gu.stream())
    qkv = F.layer_norm(x, (Cc,)) @ w.float().T
    q = bf(qkv[:, :Cc]).view(B, N, Cc).contiguous(); k = bf(qkv[:, Cc:2 * Cc]).view(B, N, Cc).contiguous()
    vt = bf(qkv[:, 2 * Cc:]).view(B, N, Cc).transpose(1, 2).contiguous()
    o = torch.empty(B, N, Cc, dtype=torch.bfloat16, device=DEV)
    args = (_lib.ptr(q), Cc, _lib.ptr(k), Cc, _lib.ptr(vt), N)
    f16 = lambda: _lib.call("dfh_attention", *args, _lib.ptr(o), Cc, B, H, D, N, N, D ** -0.5, gu.stream())
    f8 = lambda: _lib.call("dfh_attention_fp8", *args, _lib.ptr(o), Cc, _lib.ptr(rq), _lib.ptr(rk), _lib.ptr(rv), _lib.ptr(hs), B, H, D, N, N, D ** -0.5, gu.stream())
    res = {"bf16": [], "fp8": []}
    for rnd_i in range(6):
        for name, fn in (("bf16", f16), ("fp8", f8)):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) * 100)
    fl = 4.0 * B * H * N * N * D
    print(f"d={D:3d} N={N:4d}: bf16 {min(res['bf16']):7.1f} us ({fl / min(res['bf16']) / 1e6:6.1f} TFLOP/s)   fp8 {min(res['fp8']):7.1f} us ({fl / min(res['fp8']) / 1e6:6.1f} TFLOP/s)")
