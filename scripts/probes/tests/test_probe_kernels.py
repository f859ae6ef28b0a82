"""Parity of the probe-only kernels (scripts/probes/kernels): run with the probe library loaded,
    make -C scripts/probes && DFH_LIB=scripts/probes/build/libdifashion_probes.so python -m pytest scripts/probes/tests -q
Not part of tests/ (the product library does not contain these kernels)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from difashion_amd import _lib  # noqa: E402
from tests import gpu_util as gu  # noqa: E402
from tests.gpu_util import DEV, bf, rnd  # noqa: E402


pytestmark = pytest.mark.skipif("probes" not in os.environ.get("DFH_LIB", ""), reason="needs DFH_LIB=<probe library>")


@pytest.mark.parametrize("cin,cout,H,W,B", [
    (64, 160, 16, 16, 3),     # 16-wide level: one tile = one whole image, the patch is the zero-padded image
    (96, 200, 32, 32, 2),     # 32-wide: 8 rows per tile; C_out not a multiple of the 160-column tile
    (32, 64, 8, 64, 2),       # 64-wide: 4 rows per tile, 2 tiles per image (top and bottom borders in different tiles)
    (64, 320, 64, 64, 1),     # the 64x64 level itself: interior tiles with halo rows on both sides
    (160, 96, 24, 32, 2),     # height not a power of two
])
def test_conv3x3_halo_patch_kernel(cin, cout, H, W, B):
    """gemm_halo.hip (tile id 20): the pixels of a channel slice are staged once as a (rows + 2) x (W + 2) patch and the nine taps
    read it at shifted offsets.  Against fp32 conv2d of the same bf16 operands, with bias, per-image time-embedding row and residual
    (the wide kernel's epilogue), and bit for bit against the wide kernel (same MFMA order of accumulation per output)."""
    x = bf(rnd(B, cin, H, W, seed=26))
    w = rnd(cout, cin, 3, 3, seed=27, scale=0.05)
    bias = rnd(cout, seed=28)
    temb = rnd(B, 2 * cout, seed=29)
    res = bf(rnd(B * H * W, cout, seed=30))
    kw = dict(M=B * H * W, N=cout, W=gu.pack_conv(w), ldw=9 * cin, conv_src=gu.nhwc(x), conv_c=cin, batch=B, Hin=H, Win=W, stride=1,
              upsample=0, bias=bias, rowvec=temb, rv_ld=2 * cout, rv_off=cout, rows_per_b=H * W, resid=res)
    out = gu.gemm(force_tile=20, **kw)
    ref = F.conv2d(x.float(), bf(w).float(), bias, padding=1) + temb[:, cout:, None, None]
    ref = ref + gu.nchw(res.float().view(B, H, W, cout))
    gu.assert_close_bf16(gu.nchw(out.view(B, H, W, cout)), ref, f"halo conv {cin}->{cout}@{H}x{W}")
    wide = gu.gemm(force_tile=6, **kw)
    assert torch.equal(out, wide), "halo and wide kernels accumulate the same products in the same order"


@pytest.mark.parametrize("tile", [7, 8, 11, 12])    # 128-row wide sibling, eight-wave 256 x 320, wave-specialised 160 / 128
@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (128, 160, 64), (16, 1280, 200)])
def test_probe_gemm_tiles(tile, M, N, K):
    a, w = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=0.05))
    bias = rnd(N, seed=3)
    res = bf(rnd(M, N, seed=4))
    out = gu.gemm(M=M, N=N, W=w, ldw=K, a0=a, a0_c=K, bias=bias, resid=res, force_tile=tile)
    gu.assert_close_bf16(out, a.float() @ w.float().T + bias + res.float(), f"probe tile {tile}")


@pytest.mark.parametrize("M,C,fold", [(8192, 512, False), (16384, 320, True), (4096, 1280, True), (8192, 512, True)])
def test_geglu_persistent_rows_kernel_is_bit_identical(M, C, fold, monkeypatch):
    """gemm_geglu_rows_kernel (csrc/gemm_geglu.hip: one workgroup per CU walking 256 x 256 tiles, the next tile's first stage fetched
    during the current tile's epilogue) against the one-tile-per-workgroup kernel it specialises (tile id 23): same staging, MFMA order
    and epilogue arithmetic -> bit-identical, with plain bias and with the folded-LayerNorm fix-up; and against torch."""
    import ctypes
    monkeypatch.setenv("DFH_GEGLU_ROWS", "1")            # opt-in kernel (measured equal inside the step: csrc/gemm_geglu.hip)
    x = bf(rnd(M, C, seed=61) + (2.0 if fold else 0.0))
    w = rnd(8 * C, C, seed=62, scale=0.05)
    b = rnd(8 * C, seed=63, scale=0.5)
    wp = torch.empty((8 * C, C), dtype=torch.bfloat16, device=DEV)
    bp = torch.empty(8 * C, dtype=torch.float32, device=DEV)
    _lib.call("dfh_pack_matrix", _lib.ptr(w), _lib.ptr(wp), 8 * C, C, C, 0, 0, 1, gu.stream())
    _lib.call("dfh_pack_vector", _lib.ptr(b), _lib.ptr(bp), 8 * C, 0, 1, 0, gu.stream())
    assert (M // 256) * (8 * C // 256) >= 512            # enough tiles for the persistent kernel to take the launch
    if not fold:
        _lib.census_reset()
        out = gu.gemm(M=M, N=8 * C, W=wp, ldw=C, a0=x, a0_c=C, bias=bp, act=4)
        assert _lib.census()["gemm_rows_geglu"] == 1
        one = gu.gemm(M=M, N=8 * C, W=wp, ldw=C, a0=x, a0_c=C, bias=bp, act=4, force_tile=23)
        h = x.float() @ bf(w).float().T + b
    else:
        gamma, beta = 1.0 + 0.2 * rnd(C, seed=64), 0.3 * rnd(C, seed=65)
        wf = torch.empty_like(wp)
        sv, bv = torch.empty(8 * C, device=DEV), torch.empty(8 * C, device=DEV)
        _lib.call("dfh_ln_fold", _lib.ptr(wp), C, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(bp), _lib.ptr(wf), _lib.ptr(sv), _lib.ptr(bv), 8 * C, C,
                  gu.stream())
        xf = x.float()
        mean = xf.mean(-1)
        st = torch.stack([mean, ((xf - mean[:, None]) ** 2).sum(-1)], -1).contiguous()       # one column tile of C columns: [1][M][2]
        outs = []
        for tile in (0, 23):
            d = gu.gemm_desc(M=M, N=8 * C, W=wf, ldw=C, a0=x, a0_c=C, bias=bv, act=4, force_tile=tile)
            _lib.census_reset()
            _lib.call("dfh_gemm_ln", ctypes.byref(d), None, None, _lib.ptr(st), 1, C, 1e-5, _lib.ptr(sv), gu.stream())
            torch.cuda.synchronize()
            assert _lib.census()["gemm_rows_geglu"] == (1 if tile == 0 else 0)
            outs.append(d.keep_out)
        out, one = outs
        h = F.layer_norm(xf, (C,), gamma, beta, 1e-5) @ bf(w).float().T + b
    assert torch.equal(out, one), "persistent kernel differs from the one-tile-per-workgroup kernel"
    a, gate = h.chunk(2, -1)
    gu.assert_close_bf16(out, a * F.gelu(gate), "geglu rows", rel=8e-3 if fold else 6e-3)


@pytest.mark.parametrize("M,resid,folded", [(128, False, False), (640, True, False), (1024, False, True), (384, False, True)])   # folded consumers never carry a residual (gemm_ln_consumer_ok)
def test_token_linear_matches_torch(M, resid, folded):
    """token_linear.hip (tile id 30 of dfh_gemm / dfh_gemm_ln): the K = N = 320 projections of the 64x64-level transformer blocks with the
    rows held in registers -- plain (+ bias, + residual) and as a folded-LayerNorm consumer -- against fp32 torch on the same bf16 operands
    and against the tile GEMM; the per-row statistics it leaves for the next folded consumer (ONE record per row over all 320 columns)
    against torch on its own rounded output."""
    import ctypes
    C = 320
    x = bf(rnd(M, C, seed=81) + 1.5)
    res = bf(rnd(M, C, seed=82) * 2.0) if resid else None
    w = bf(rnd(C, C, seed=83, scale=0.05))
    bias = rnd(C, seed=84, scale=0.3)
    rs = torch.full((M, 2), float("nan"), device=DEV)
    bn = ctypes.c_int(0)
    if not folded:
        d = gu.gemm_desc(M=M, N=C, W=w, ldw=C, a0=x, a0_c=C, bias=bias, resid=res, force_tile=30)
        _lib.call("dfh_gemm_ln", ctypes.byref(d), _lib.ptr(rs), ctypes.byref(bn), None, 0, 0, 0.0, None, gu.stream())
        ref = x.float() @ w.float().T + bias + (res.float() if resid else 0)
    else:
        gamma, beta = 1.0 + 0.2 * rnd(C, seed=85), 0.3 * rnd(C, seed=86)
        wf = torch.empty_like(w); s1, b1 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        _lib.call("dfh_ln_fold", _lib.ptr(w), C, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(bias), _lib.ptr(wf), _lib.ptr(s1), _lib.ptr(b1), C, C, gu.stream())
        mean = x.float().mean(-1)
        st = torch.stack([mean, ((x.float() - mean[:, None]) ** 2).sum(-1)], dim=-1).contiguous()       # one record per row
        d = gu.gemm_desc(M=M, N=C, W=wf, ldw=C, a0=x, a0_c=C, bias=b1, resid=res, force_tile=30)
        _lib.call("dfh_gemm_ln", ctypes.byref(d), _lib.ptr(rs), ctypes.byref(bn), _lib.ptr(st), 1, C, 1e-5, _lib.ptr(s1), gu.stream())
        ref = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5) @ w.float().T + bias + (res.float() if resid else 0)
    torch.cuda.synchronize()
    out = d.keep_out
    assert bn.value == C
    gu.assert_close_bf16(out, ref, "token linear")
    o = out.float()
    torch.testing.assert_close(rs[:, 0], o.mean(-1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(rs[:, 1], ((o - o.mean(-1, keepdim=True)) ** 2).sum(-1), rtol=1e-4, atol=1e-4)
    if not folded:
        plain = gu.gemm(M=M, N=C, W=w, ldw=C, a0=x, a0_c=C, bias=bias, resid=res)
        assert gu.rel_err(out, plain) < 4e-3


def test_attention_backward_x32_variant_is_parity_green():
    """attention_bwd.hip X32 (probe builds, DFH_ATTN_BWD_X32=1): S / dP of the d = 40 backward on 32x32x16 with the P / dS re-layout through
    v_permlane16_swap.  Slower than the product path (profiles/r05/attn_bwd_x32_ab.txt) but it must stay correct: the product's own
    attention-backward test, run in a child process with the variant switched on (the switch is read once per process)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    env = dict(os.environ, DFH_ATTN_BWD_X32="1")
    assert "probes" in env.get("DFH_LIB", ""), "run with DFH_LIB=<probe library>"
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_backward.py", "-x", "-q", "-k", "attention_backward and 40-8"],
                       capture_output=True, text=True, env=env, cwd=root, timeout=900)
    assert r.returncode == 0 and "5 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


PERSIST_CASES = {
    # name: (M, N, K segments, options)   -- whole 128 x 160 tiles, K >= 256 in 64-deep steps (gemm_persist_ok)
    "resid_two_tiles_per_cu": (128 * 600, 320, (320,), dict(bias=True, resid=True)),               # 1200 tiles: every workgroup walks 4-5 tiles
    "plain_ragged_rounds": (128 * 301, 320, (320,), dict(bias=False, resid=False)),                # 602 tiles over 256 workgroups: 2 or 3 each
    "single_round": (128 * 9, 160, (256,), dict(bias=True, resid=True)),                           # 9 tiles: one per workgroup, no ring reuse
    "silu_k640": (128 * 160, 640, (640,), dict(bias=True, resid=True, act=1)),
    "two_segments": (128 * 130, 320, (1280, 320), dict(bias=True, resid=True)),                    # ff.net.2 . proj_out: [hidden | x] operands
    "transposed": (6 * 1024, 320, (320,), dict(bias=True, out_t=True, rows_per_b=1024)),           # V^T
    "qk_and_vt": (6 * 4096, 960, (320,), dict(out2=True, rows_per_b=4096)),                        # q | k | V^T from one launch
    "rowstat_producer": (128 * 300, 640, (320,), dict(bias=True, resid=True, rowstat=True)),
    "ln_consumer": (128 * 300, 640, (320,), dict(bias=True, ln_parts=2)),
    "ln_consumer_4_parts": (128 * 150, 960, (640,), dict(bias=False, ln_parts=4)),
    "gstat_32x32": (24 * 1024, 640, (640,), dict(bias=True, resid=True, gstat=(20, 1024))),
    "per_image_weights": (6 * 4096, 320, (320,), dict(rowvec_img=True, w_img=True, rows_per_b=4096)),   # GroupNorm folded into proj_in
}


@pytest.mark.parametrize("name", list(PERSIST_CASES))
def test_gemm_persistent_matches_the_tile_kernel_bit_for_bit(name):
    """gemm_persist.hip (tile id 24: one workgroup per CU walks its 128 x 160 tiles, the LDS ring runs across tiles, every epilogue operand
    arrives by LDS-DMA, counted waits only) against gemm_bf16_kernel<128,160,4,2,*,LEAN> (tile id 10) on the same operands: the same
    MFMA sequence per tile and the same epilogue arithmetic -> every output, row-statistics record and GroupNorm partial BIT for bit, and
    the fp32 reference within the bf16 tolerance.  Runs each launch twice (a race on the ring or the staging region would differ)."""
    import ctypes
    M, N, segs, o = PERSIST_CASES[name]
    K = sum(segs)
    a0 = bf(rnd(M, segs[0], seed=201))
    a1 = bf(rnd(M, segs[1], seed=202)) if len(segs) > 1 else None
    rows_per_b = o.get("rows_per_b", 0)
    B = M // rows_per_b if rows_per_b else 1
    w = bf(rnd((B if o.get("w_img") else 1) * N, K, seed=203, scale=0.05))
    bias = rnd(N, seed=204) if o.get("bias") else None
    resid = bf(rnd(M, N, seed=205)) if o.get("resid") else None
    rowvec = rnd(B, N, seed=206) if o.get("rowvec_img") else None
    parts = o.get("ln_parts", 0)
    ln_stat = ln_s = None
    if parts:
        cnt = segs[0] // parts
        xp = a0.float().view(M, parts, cnt).transpose(0, 1)
        mean_t = xp.mean(-1)
        ln_stat = torch.stack([mean_t, ((xp - mean_t[..., None]) ** 2).sum(-1)], dim=-1).contiguous()      # [parts][M][2]
        ln_s = w.float().sum(1).contiguous()

    def run(tile):
        kw = dict(M=M, N=N, W=w, ldw=K, a0=a0, a0_c=segs[0], bias=bias, resid=resid, act=o.get("act", 0), force_tile=tile,
                  rows_per_b=rows_per_b)
        if a1 is not None:
            kw.update(a1=a1, a1_c=segs[1])
        if rowvec is not None:
            kw.update(rowvec=rowvec, rv_ld=N, rv_off=0)
        if o.get("w_img"):
            kw.update(w_img_stride=N * K)
        res = {}
        if o.get("out_t"):
            out = torch.zeros((B, N, rows_per_b + 8), dtype=torch.bfloat16, device=DEV)
            kw.update(out=out, ld_out=rows_per_b + 8, out_mode=1)
        if o.get("out2"):
            qk = torch.zeros((M, 640), dtype=torch.bfloat16, device=DEV)
            vt = torch.zeros((B, N - 640, rows_per_b + 8), dtype=torch.bfloat16, device=DEV)
            d = gu.gemm_desc(**dict(kw, out=qk, ld_out=640))
            _lib.call("dfh_gemm_out2", ctypes.byref(d), _lib.ptr(vt), rows_per_b + 8, 640, gu.stream())
            torch.cuda.synchronize()
            return dict(out=qk, vt=vt)
        if o.get("rowstat") or parts:
            d = gu.gemm_desc(**kw)
            st = torch.full((8 * M * 2,), float("nan"), dtype=torch.float32, device=DEV) if o.get("rowstat") else None
            bn = ctypes.c_int(0)
            _lib.call("dfh_gemm_ln", ctypes.byref(d), _lib.ptr(st), ctypes.byref(bn) if st is not None else None,
                      _lib.ptr(ln_stat), parts, (segs[0] // parts) if parts else 0, 1e-5, _lib.ptr(ln_s), gu.stream())
            torch.cuda.synchronize()
            res["out"] = d.keep_out
            if st is not None:
                assert bn.value == 160
                res["rowstat"] = st[:(N // 160) * M * 2].clone()
            return res
        if o.get("gstat"):
            cpg, hw = o["gstat"]
            gst = torch.full(((M // hw) * (N // cpg) * (hw // 128) * 2,), float("nan"), dtype=torch.float32, device=DEV)
            out, rows = gu.gemm(gstat=gst, gstat_cpg=cpg, gstat_hw=hw, **kw)
            assert rows == 128
            return dict(out=out, gstat=gst)
        return dict(out=gu.gemm(**kw))

    _lib.census_reset()
    got = run(24)
    assert _lib.census()["gemm_persist"] == 1
    again = run(24)
    ref = run(10)
    assert _lib.census()["gemm_persist"] == 2
    for k in ref:
        assert torch.equal(got[k], ref[k]), (name, k, float((got[k].float() - ref[k].float()).abs().max()))
        assert torch.equal(again[k], got[k]), (name, k, "rerun differs")
    # ... and the tile kernel itself is held to fp32 by the tests above; one direct check here for the plain shapes
    if not parts and not o.get("out_t") and not o.get("out2") and not o.get("w_img") and o.get("act", 0) == 0:
        A = a0.float() if a1 is None else torch.cat([a0.float(), a1.float()], 1)
        want = A @ w.float().T + (bias if bias is not None else 0) + (resid.float() if resid is not None else 0)
        gu.assert_close_bf16(got["out"], want, name)
