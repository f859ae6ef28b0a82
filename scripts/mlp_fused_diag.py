#!/usr/bin/env python3
"""Diagnosis of dfh_mlp_fused: which term / which output positions deviate from fp32 torch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch.nn.functional as F
from difashion_amd import _lib
import gpu_util as gu
from gpu_util import bf, rnd
DEV = "cuda"
FORM = int(os.environ.get("FORM", 2))
C, M = 320, 256

def run(tag, zero_hidden=False, zero_wp=False, zero_bias1=False, unit_ln=False, offset=0.0, wscale=0.05):
    x = bf(rnd(M, C, seed=71) + offset)
    resid = bf(torch.zeros(M, C, device=DEV))
    gamma, beta = (torch.ones(C, device=DEV), torch.zeros(C, device=DEV)) if unit_ln else (1.0 + 0.2 * rnd(C, seed=73), 0.3 * rnd(C, seed=74))
    wg, bg = rnd(8 * C, C, seed=75, scale=wscale), rnd(8 * C, seed=76, scale=0.0 if zero_bias1 else 0.3)
    wp = torch.empty((8 * C, C), dtype=torch.bfloat16, device=DEV); bp = torch.empty(8 * C, dtype=torch.float32, device=DEV)
    _lib.call("dfh_pack_matrix", _lib.ptr(wg), _lib.ptr(wp), 8 * C, C, C, 0, 0, 1, gu.stream())
    _lib.call("dfh_pack_vector", _lib.ptr(bg), _lib.ptr(bp), 8 * C, 0, 1, 0, gu.stream())
    wf = torch.empty_like(wp); s1, b1 = torch.empty(8 * C, device=DEV), torch.empty(8 * C, device=DEV)
    _lib.call("dfh_ln_fold", _lib.ptr(wp), C, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(bp), _lib.ptr(wf), _lib.ptr(s1), _lib.ptr(b1), 8 * C, C, gu.stream())
    w2p = bf(torch.cat([rnd(C, 4 * C, seed=77, scale=0.0 if zero_hidden else 0.03), rnd(C, C, seed=79, scale=0.0 if zero_wp else 0.05)], dim=1)).contiguous()
    bias = torch.zeros(C, device=DEV)
    parts, cnt = 2, 160
    xp = x.float().view(M, parts, cnt).transpose(0, 1); mean_t = xp.mean(-1)
    st = torch.stack([mean_t, ((xp - mean_t[..., None]) ** 2).sum(-1)], dim=-1).contiguous()
    img = torch.empty(_lib.raw().dfh_mlp_fused_image_bytes(), dtype=torch.uint8, device=DEV)
    _lib.call("dfh_mlp_fused_pack", _lib.ptr(wf), _lib.ptr(s1), _lib.ptr(b1), _lib.ptr(w2p), _lib.ptr(img), FORM, gu.stream())
    out = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=DEV)
    _lib.call("dfh_mlp_fused", _lib.ptr(x), _lib.ptr(resid), _lib.ptr(img), _lib.ptr(st), parts, cnt, 1e-5, _lib.ptr(bias), _lib.ptr(out), M, FORM, None, 0, 0, gu.stream())
    torch.cuda.synchronize()
    ln = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5)
    hh = ln @ bf(wg).float().T + bg
    av, gate = hh.chunk(2, -1)
    hid = bf(av * F.gelu(gate)).float()
    ref = torch.cat([hid, x.float()], dim=1) @ w2p.float().T
    d = (out.float() - ref)
    rel = float(d.norm() / ref.norm())
    # error energy by output channel tile (32) and by token within its wave (32)
    ct = (d ** 2).view(M, 10, 32).sum((0, 2)).sqrt() / ((ref ** 2).view(M, 10, 32).sum((0, 2)).sqrt() + 1e-9)
    tk = (d ** 2).view(M // 32, 32, C).sum((0, 2)).sqrt() / ((ref ** 2).view(M // 32, 32, C).sum((0, 2)).sqrt() + 1e-9)
    print(f"{tag:34s} rel {rel:.3e} | by channel tile {[f'{v:.1e}' for v in ct.tolist()]} | token max/min {float(tk.max()):.1e}/{float(tk.min()):.1e}", flush=True)

run("all terms")
run("h2 segment only (hidden weights 0)", zero_hidden=True)
run("hidden only (pout part 0)", zero_wp=True)
run("hidden only, unit LayerNorm", zero_wp=True, unit_ln=True)
run("hidden only, unit LN, no bias1", zero_wp=True, unit_ln=True, zero_bias1=True)
run("hidden only, offset 3", zero_wp=True, offset=3.0)
run("hidden only, small W1 (gelu ~ linear)", zero_wp=True, wscale=0.005, zero_bias1=True, unit_ln=True)
