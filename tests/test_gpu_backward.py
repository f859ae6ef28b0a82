"""-m gpu: backward-pass kernels (training step) through the C ABI against
torch autograd in fp32 on the same bf16-rounded operands."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from difashion_amd import _lib
from tests import gpu_util as gu
from tests.gpu_util import DEV, bf, rnd

pytestmark = pytest.mark.gpu


def wgrad(*, M, N, dY, conv_src=None, conv_c=0, batch=0, Hin=0, stride=1, upsample=0, a0=None, a0_c=0, a1=None, a1_c=0,
          ldw, msplit=0, into=None):
    d = _lib.GemmDesc()
    if conv_src is not None:
        d.conv_src, d.conv_c, d.conv = conv_src.data_ptr(), conv_c, 1
        d.batch, d.Hin, d.Win, d.stride, d.upsample = batch, Hin, Hin, stride, upsample
    if a0 is not None:
        d.a0, d.a0_c = a0.data_ptr(), a0_c
    if a1 is not None:
        d.a1, d.a1_c = a1.data_ptr(), a1_c
    d.M, d.N = M, N
    z = gu.zero_page()
    d.zero_page = z.data_ptr()
    dW = torch.zeros((N, ldw), dtype=torch.float32, device=DEV) if into is None else into
    need = _lib.raw().dfh_gemm_wgrad_partial_floats(C.byref(d), msplit)
    part = torch.empty(max(need, 1), dtype=torch.float32, device=DEV)
    d.partial, d.partial_floats = part.data_ptr(), need
    _lib.call("dfh_gemm_wgrad", C.byref(d), _lib.ptr(dY), dY.shape[-1], _lib.ptr(dW), ldw, msplit, gu.stream())
    torch.cuda.synchronize()
    return dW


@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (4096, 160, 64), (77 * 3, 64, 768), (130, 1280, 200), (8, 256, 2048)])
@pytest.mark.parametrize("msplit", [0, 1, 3, 7, -4])
def test_wgrad_linear(M, N, K, msplit):
    a, dy = bf(rnd(M, K, seed=1)), bf(rnd(M, N, seed=2))
    dW = wgrad(M=M, N=N, dY=dy, a0=a, a0_c=K, ldw=K, msplit=msplit)
    ref = dy.float().T @ a.float()
    assert gu.rel_err(dW, ref) < 2e-5, gu.rel_err(dW, ref)      # fp32 accumulate of exact bf16 products


@pytest.mark.parametrize("M,N,K,msplit", [(2048, 1280, 11520, -8),      # 576 tiles: 512 whole-tile blocks + the last 64 tiles as 8 pixel slices each
                                          (256, 1280, 23040, -4),       # 1152 tiles: two full rounds of whole tiles + 128 tiles x 4 slices
                                          (2048, 1280, 11520, 0),       # conv 1280->1280 at 8x8 without the conv addressing: the plan's own choice
                                          (9000, 320, 2880, 14), (9000, 320, 2880, 13)])   # slice counts that are not multiples of 8
def test_wgrad_piece_decompositions(M, N, K, msplit):
    """The launch as pieces (wgrad.hip): equal pixel slices in the XCD-contiguous block order for ANY slice count, and the whole / tail
    split (full rounds of whole-tile blocks that add straight into dW, the remaining tiles sliced); the slab reduce must cover exactly the
    sliced tiles.  Accumulating launch (dW += ...) on top of a non-zero dW, so a piece added twice or never shows; bit-identical reruns."""
    a, dy = bf(rnd(M, K, seed=21)), bf(rnd(M, N, seed=22))
    ref = dy.float().T @ a.float()
    dW = wgrad(M=M, N=N, dY=dy, a0=a, a0_c=K, ldw=K, msplit=msplit)
    assert gu.rel_err(dW, ref) < 2e-5, gu.rel_err(dW, ref)
    dW2 = wgrad(M=M, N=N, dY=dy, a0=a, a0_c=K, ldw=K, msplit=msplit, into=dW.clone())
    assert gu.rel_err(dW2, 2 * ref) < 2e-5
    assert torch.equal(wgrad(M=M, N=N, dY=dy, a0=a, a0_c=K, ldw=K, msplit=msplit), dW), "the piece order of the slab reduce is fixed: bit-identical reruns"


def test_wgrad_two_sources_and_column_offset():
    """concat input (up-path shortcut): dW columns [0:C0] from a0, [C0:C0+C1] from a1, inside a wider packed matrix"""
    M, N, C0, C1 = 512, 96, 192, 64
    a0, a1, dy = bf(rnd(M, C0, seed=3)), bf(rnd(M, C1, seed=4)), bf(rnd(M, N, seed=5))
    dW = wgrad(M=M, N=N, dY=dy, a0=a0, a0_c=C0, a1=a1, a1_c=C1, ldw=C0 + C1)
    ref = dy.float().T @ torch.cat([a0, a1], 1).float()
    assert gu.rel_err(dW, ref) < 2e-5


@pytest.mark.parametrize("cin,cout,H,stride,ups", [(64, 160, 16, 1, 0), (32, 64, 8, 1, 0), (320, 64, 16, 1, 0), (64, 64, 16, 2, 0),
                                                   (64, 128, 8, 1, 1), (8, 64, 16, 1, 0), (128, 128, 2, 1, 0)])
@pytest.mark.parametrize("msplit", [0, 5, -3])
def test_wgrad_conv3x3(cin, cout, H, stride, ups, msplit):
    B = 2
    x = bf(rnd(B, cin, H, H, seed=6))
    Ho = H * 2 if ups else H // stride
    dy = bf(rnd(B, cout, Ho, Ho, seed=7))
    dW = wgrad(M=B * Ho * Ho, N=cout, dY=gu.nhwc(dy).view(-1, cout), conv_src=gu.nhwc(x), conv_c=cin, batch=B, Hin=H,
               stride=stride, upsample=ups, ldw=9 * cin, msplit=msplit)
    w = torch.zeros(cout, cin, 3, 3, device=DEV, requires_grad=True)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
    y = F.conv2d(xin, w, None, stride=stride, padding=1)
    (g,) = torch.autograd.grad(y, w, dy.float())
    ref = g.permute(0, 2, 3, 1).reshape(cout, 9 * cin)
    assert gu.rel_err(dW, ref) < 2e-5, gu.rel_err(dW, ref)


def test_wgrad_conv_plus_shortcut_segments():
    """resnet tail: conv2 taps over h followed by the 1x1 shortcut segments over (x0, x1), one packed matrix"""
    B, H, c0, c1, cout = 2, 8, 64, 32, 96
    h, x0, x1 = bf(rnd(B, cout, H, H, seed=8)), bf(rnd(B, c0, H, H, seed=9)), bf(rnd(B, c1, H, H, seed=10))
    dy = bf(rnd(B, cout, H, H, seed=11))
    M, K = B * H * H, 9 * cout + c0 + c1
    s0, s1 = gu.nhwc(x0).view(M, c0), gu.nhwc(x1).view(M, c1)
    dW = wgrad(M=M, N=cout, dY=gu.nhwc(dy).view(M, cout), conv_src=gu.nhwc(h), conv_c=cout, batch=B, Hin=H,
               a0=s0, a0_c=c0, a1=s1, a1_c=c1, ldw=K)
    w2 = torch.zeros(cout, cout, 3, 3, device=DEV, requires_grad=True)
    ws = torch.zeros(cout, c0 + c1, 1, 1, device=DEV, requires_grad=True)
    y = F.conv2d(h.float(), w2, padding=1) + F.conv2d(torch.cat([x0, x1], 1).float(), ws)
    g2, gs = torch.autograd.grad(y, (w2, ws), dy.float())
    ref = torch.cat([g2.permute(0, 2, 3, 1).reshape(cout, -1), gs.reshape(cout, -1)], 1)
    assert gu.rel_err(dW, ref) < 2e-5


def test_colsum_bias_and_time_embedding_grads():
    B, HW, N = 3, 200, 320
    dy = bf(rnd(B * HW, N, seed=12))
    out = torch.zeros(1, N, device=DEV)
    _lib.call("dfh_colsum", _lib.ptr(dy), N, N, 1, B * HW, _lib.ptr(out), N, gu.stream())
    outb = torch.zeros(B, 2 * N, device=DEV)
    _lib.call("dfh_colsum", _lib.ptr(dy), N, N, B, HW, _lib.ptr(outb[:, N:]), 2 * N, gu.stream())
    torch.cuda.synchronize()
    assert gu.rel_err(out[0], dy.float().sum(0)) < 1e-5
    assert gu.rel_err(outb[:, N:], dy.float().view(B, HW, N).sum(1)) < 1e-5
    assert float(outb[:, :N].abs().max()) == 0.0


# ----------------------------------------------------------------------------- data gradients through the forward GEMM
def test_dgrad_linear_with_transposed_pack():
    M, N, K = 300, 320, 192
    w = rnd(N, K, seed=20, scale=0.05)
    dy = bf(rnd(M, N, seed=21))
    wt = torch.zeros((K, N), dtype=torch.bfloat16, device=DEV)
    _lib.call("dfh_pack_matrix_t", _lib.ptr(w), _lib.ptr(wt), N, K, N, 0, 0, 0, gu.stream())
    dx = gu.gemm(M=M, N=K, W=wt, ldw=N, a0=dy, a0_c=N)
    gu.assert_close_bf16(dx, dy.float() @ bf(w).float(), "dgrad linear")


@pytest.mark.parametrize("cin,cout,H,stride,ups", [(64, 96, 16, 1, 0), (32, 64, 8, 1, 0), (64, 64, 16, 2, 0), (64, 128, 8, 1, 1),
                                                   (128, 128, 2, 1, 0)])
def test_dgrad_conv3x3(cin, cout, H, stride, ups):
    """dX = conv3x3(dY, flipped/transposed W): stride 1 directly, stride 2 over the zero-inserted dY (upsample = 2),
    nearest-2x upsample + conv as dgrad at 2H followed by the 2x2 sum pool."""
    B = 2
    x = bf(rnd(B, cin, H, H, seed=22)).float().requires_grad_(True)
    w = rnd(cout, cin, 3, 3, seed=23, scale=0.05)
    Ho = H * 2 if ups else H // stride
    dy = bf(rnd(B, cout, Ho, Ho, seed=24))
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    y = F.conv2d(xin, bf(w).float(), None, stride=stride, padding=1)
    (ref,) = torch.autograd.grad(y, x, dy.float())
    wt = torch.zeros((cin, 9 * cout), dtype=torch.bfloat16, device=DEV)
    _lib.call("dfh_pack_conv3x3_t", _lib.ptr(w), _lib.ptr(wt), cout, cin, 9 * cout, 0, cout, gu.stream())
    dyn = gu.nhwc(dy)
    if stride == 2:
        dx = gu.gemm(M=B * H * H, N=cin, W=wt, ldw=9 * cout, conv_src=dyn, conv_c=cout, batch=B, Hin=Ho, Win=Ho, upsample=2)
    else:
        dx = gu.gemm(M=B * Ho * Ho, N=cin, W=wt, ldw=9 * cout, conv_src=dyn, conv_c=cout, batch=B, Hin=Ho, Win=Ho)
        if ups:
            pooled = torch.empty((B * H * H, cin), dtype=torch.bfloat16, device=DEV)
            _lib.call("dfh_pool2x2_sum", _lib.ptr(dx), _lib.ptr(pooled), B, H, H, cin, gu.stream())
            torch.cuda.synchronize()
            dx = pooled
    gu.assert_close_bf16(gu.nchw(dx.view(B, H, H, cin)), ref, f"dgrad conv s{stride} u{ups}", rel=8e-3, max_rel=4e-2)


# ----------------------------------------------------------------------------- normalisation backward
@pytest.mark.parametrize("C0,C1,HW,silu,eps", [(320, 0, 256, 1, 1e-5), (640, 320, 64, 1, 1e-5), (64, 0, 4, 0, 1e-6), (32, 32, 256, 1, 1e-5)])
def test_groupnorm_backward(C0, C1, HW, silu, eps):
    import math
    B, G = 3, 32
    C, H = C0 + C1, int(math.isqrt(HW))
    x0 = bf(rnd(B, C0, H, H, seed=31) * 2 + 0.7)
    x1 = bf(rnd(B, C1, H, H, seed=32) - 0.4) if C1 else None
    gamma, beta = rnd(C, seed=33) * 0.3 + 1, rnd(C, seed=34) * 0.2
    dy = bf(rnd(B, C, H, H, seed=35))
    s0, s1, dyn = gu.nhwc(x0), (gu.nhwc(x1) if C1 else None), gu.nhwc(dy)
    out = torch.empty((B, HW, C), dtype=torch.bfloat16, device=DEV)
    part = torch.empty(B * 64 * G * 2, device=DEV)
    stats = torch.empty(B * G * 2, device=DEV)
    _lib.call("dfh_groupnorm_stats", _lib.ptr(s0), C0, _lib.ptr(s1), C1, B, HW, G, _lib.ptr(gamma), _lib.ptr(beta), eps, silu,
              _lib.ptr(out), _lib.ptr(part), _lib.ptr(stats), gu.stream())
    pre0 = bf(rnd(B, HW, C0, seed=36))                   # existing gradient contents to accumulate into
    dx0, dx1 = pre0.clone(), (torch.empty((B, HW, C1), dtype=torch.bfloat16, device=DEV) if C1 else None)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    _lib.call("dfh_groupnorm_bwd", _lib.ptr(s0), C0, _lib.ptr(s1), C1, _lib.ptr(dyn), B, HW, G, _lib.ptr(gamma), _lib.ptr(beta),
              _lib.ptr(stats), silu, _lib.ptr(dx0), 1, _lib.ptr(dx1), 0, _lib.ptr(dg), _lib.ptr(db), _lib.ptr(part), gu.stream())
    torch.cuda.synchronize()
    xin = (torch.cat([x0, x1], 1) if C1 else x0).float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = F.group_norm(xin, G, gr, br, eps)
    y = F.silu(y) if silu else y
    rx, rg, rb = torch.autograd.grad(y, (xin, gr, br), dy.float())
    got0 = gu.nchw(dx0.view(B, H, H, C0)).float() - gu.nchw(pre0.view(B, H, H, C0)).float()
    gu.assert_close_bf16(got0, rx[:, :C0], "gn dx0 (accumulated)", rel=2e-2, max_rel=5e-2)
    if C1:
        gu.assert_close_bf16(gu.nchw(dx1.view(B, H, H, C1)), rx[:, C0:], "gn dx1", rel=8e-3, max_rel=4e-2)
    assert gu.rel_err(dg, rg) < 2e-3 and gu.rel_err(db, rb) < 2e-3


@pytest.mark.parametrize("M,C", [(300, 320), (64, 1280), (37, 32), (130, 640)])
def test_layernorm_backward(M, C):
    x = bf(rnd(M, C, seed=37) * 3 + 1.5)
    dy = bf(rnd(M, C, seed=38))
    gamma, beta = rnd(C, seed=39) * 0.3 + 1, rnd(C, seed=40) * 0.2
    dx = torch.empty_like(x)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    _lib.call("dfh_layernorm_bwd", _lib.ptr(x), _lib.ptr(dy), _lib.ptr(gamma), _lib.ptr(dx), 0, _lib.ptr(dg), _lib.ptr(db), M, C,
              1e-5, gu.stream())
    torch.cuda.synchronize()
    xr, gr, br = x.float().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rx, rg, rb = torch.autograd.grad(F.layer_norm(xr, (C,), gr, br, 1e-5), (xr, gr, br), dy.float())
    gu.assert_close_bf16(dx, rx, "ln dx", rel=8e-3, max_rel=4e-2)
    assert gu.rel_err(dg, rg) < 2e-3 and gu.rel_err(db, rb) < 2e-3


def test_geglu_forward_backward_on_packed_layout():
    M, C4 = 100, 128                      # 4C outputs, 8C packed pre-activations
    pre_std = rnd(M, 2 * C4, seed=41)     # [value | gate] standard order
    dy = bf(rnd(M, C4, seed=42))
    idx = torch.arange(2 * C4)
    j = torch.where(idx < C4, idx, idx - C4)
    packed_col = (j // 16) * 32 + torch.where(idx < C4, 0, 16) + (j % 16)
    pre = torch.empty((M, 2 * C4), dtype=torch.bfloat16, device=DEV)
    pre[:, packed_col.to(DEV)] = bf(pre_std)
    y = torch.empty((M, C4), dtype=torch.bfloat16, device=DEV)
    dpre = torch.empty_like(pre)
    _lib.call("dfh_geglu_fwd", _lib.ptr(pre), _lib.ptr(y), M, 2 * C4, gu.stream())
    _lib.call("dfh_geglu_bwd", _lib.ptr(pre), _lib.ptr(dy), _lib.ptr(dpre), M, 2 * C4, gu.stream())
    torch.cuda.synchronize()
    p = bf(pre_std).float().requires_grad_(True)
    v, g = p.chunk(2, -1)
    ref = v * F.gelu(g)
    (rp,) = torch.autograd.grad(ref, p, dy.float())
    gu.assert_close_bf16(y, ref, "geglu fwd")
    gu.assert_close_bf16(dpre[:, packed_col.to(DEV)], rp, "geglu bwd", rel=8e-3, max_rel=4e-2)


@pytest.mark.parametrize("kind,fn,from_out", [(1, F.silu, False), (2, lambda t: F.leaky_relu(t, 0.01), True), (3, torch.tanh, True)])
def test_pointwise_activation_backward(kind, fn, from_out):
    x = bf(rnd(4096, seed=43) * 2)
    dy = bf(rnd(4096, seed=44))
    xr = x.float().requires_grad_(True)
    y = fn(xr)
    (ref,) = torch.autograd.grad(y, xr, dy.float())
    refin = bf(y.detach()) if from_out else x
    out = torch.empty_like(x)
    _lib.call("dfh_act_bwd", _lib.ptr(refin), None, _lib.ptr(dy), None, _lib.ptr(out), x.numel(), kind, 1.0, gu.stream())
    torch.cuda.synchronize()
    gu.assert_close_bf16(out, ref, f"act bwd {kind}", rel=1.5e-2, max_rel=5e-2)


def test_unpack_is_inverse_of_pack_layouts():
    cout, cin = 24, 16
    g = rnd(cout, 9 * cin + 40, seed=45)                     # packed fp32 gradient [N][9*Cin | 40 more columns]
    grad = torch.zeros(cout, cin, 3, 3, device=DEV)
    _lib.call("dfh_unpack_conv3x3", _lib.ptr(g), _lib.ptr(grad), cout, cin, g.shape[1], 0, cin, gu.stream())
    gm = torch.ones(cout, 40, device=DEV)
    _lib.call("dfh_unpack_matrix", _lib.ptr(g), _lib.ptr(gm), cout, 40, g.shape[1], 0, 9 * cin, 0, gu.stream())
    torch.cuda.synchronize()
    assert torch.equal(grad, g[:, :9 * cin].view(cout, 3, 3, cin).permute(0, 3, 1, 2))
    assert torch.equal(gm, g[:, 9 * cin:] + 1.0)


def test_adamw_clip_and_ema_match_torch():
    n = 10_000
    p0, shadow0 = rnd(n, seed=46), rnd(n, seed=47)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-2, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    shadow, ref_shadow = shadow0.clone(), shadow0.clone()
    for step in range(1, 4):
        g = rnd(n, seed=50 + step) * 3
        ref_p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([ref_p], 1.0)
        opt.step()
        ref_shadow.sub_((1 - 0.999) * (ref_shadow - ref_p.detach()))
        ss = torch.zeros(1, device=DEV)
        _lib.call("dfh_sumsq", _lib.ptr(g), n, _lib.ptr(ss), gu.stream())
        _lib.call("dfh_adamw", _lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), n, 1e-2, 0.9, 0.999, 1e-8, 1e-2, step,
                  _lib.ptr(ss), 1.0, gu.stream())
        _lib.call("dfh_ema", _lib.ptr(shadow), _lib.ptr(p), n, 0.999, gu.stream())
        torch.cuda.synchronize()
        torch.testing.assert_close(ss[0], (g.double() ** 2).sum().float(), rtol=1e-5, atol=0)
        torch.testing.assert_close(p, ref_p.detach(), rtol=2e-5, atol=2e-6)
        torch.testing.assert_close(shadow, ref_shadow, rtol=1e-6, atol=1e-7)


# ----------------------------------------------------------------------------- attention backward
@pytest.mark.parametrize("d,heads", [(40, 8), (80, 4), (160, 2), (32, 2), (64, 5), (128, 2)])
@pytest.mark.parametrize("Nq,Nk", [(256, 256), (64, 64), (200, 77), (4, 4), (320, 1024)])
def test_attention_backward(d, heads, Nq, Nk):
    B, Cc = 2, d * heads
    q, k, v = bf(rnd(B, Nq, Cc, seed=60)), bf(rnd(B, Nk, Cc, seed=61)), bf(rnd(B, Nk, Cc, seed=62))
    do = bf(rnd(B, Nq, Cc, seed=63))
    ld = (Nk + 7) // 8 * 8
    vt = torch.zeros((B, Cc, ld), dtype=torch.bfloat16, device=DEV)
    vt[:, :, :Nk] = v.transpose(1, 2)
    o = torch.empty_like(q)
    lse = torch.empty((B, heads, Nq), device=DEV)
    scale = d ** -0.5
    _lib.call("dfh_attention_lse", _lib.ptr(q), Cc, _lib.ptr(k), Cc, _lib.ptr(vt), ld, _lib.ptr(o), Cc, B, heads, d, Nq, Nk,
              scale, _lib.ptr(lse), gu.stream())
    delta = torch.empty((B, heads, Nq), device=DEV)
    _lib.call("dfh_attention_delta", _lib.ptr(o), _lib.ptr(do), Cc, _lib.ptr(delta), B, heads, d, Nq, gu.stream())
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    _lib.call("dfh_attention_bwd", _lib.ptr(q), Cc, _lib.ptr(k), Cc, _lib.ptr(v), Cc, _lib.ptr(do), Cc, _lib.ptr(lse),
              _lib.ptr(delta), _lib.ptr(dq), Cc, _lib.ptr(dk), Cc, _lib.ptr(dv), Cc, B, heads, d, Nq, Nk, scale, gu.stream())
    torch.cuda.synchronize()
    qr, kr, vr = (t.float().view(B, -1, heads, d).transpose(1, 2).detach().requires_grad_(True) for t in (q, k, v))
    s = (qr @ kr.transpose(-1, -2)) * scale
    ref_o = torch.softmax(s, -1) @ vr
    ref_lse = torch.logsumexp(s, -1) * 1.4426950408889634       # log2 domain
    gq, gk, gv = torch.autograd.grad(ref_o, (qr, kr, vr), do.float().view(B, Nq, heads, d).transpose(1, 2))
    back = lambda t, n: t.transpose(1, 2).reshape(B, n, Cc)
    assert gu.max_err(lse, ref_lse) < 2e-2
    # P and dS are rounded to bf16 before their MFMAs: ~2^-8 relative on top of the bf16 output rounding
    gu.assert_close_bf16(dv, back(gv, Nk), f"dV d={d} {Nq}x{Nk}", rel=1.5e-2, max_rel=6e-2)
    gu.assert_close_bf16(dq, back(gq, Nq), f"dQ d={d} {Nq}x{Nk}", rel=1.5e-2, max_rel=6e-2)
    gu.assert_close_bf16(dk, back(gk, Nk), f"dK d={d} {Nq}x{Nk}", rel=1.5e-2, max_rel=6e-2)
