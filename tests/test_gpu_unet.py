"""-m gpu: whole-U-Net parity of the HIP path against the fp32 CPU oracle on identical weights.

Stated tolerance (north star: "within a stated fp32 tolerance"): the HIP path stores activations and
weights in bf16 (fp32 accumulate), the oracle is fp32 end to end.  Bound used here: relative L2 error of
the noise prediction <= 3e-2 and of every block-level tap <= 3e-2 (observed values are printed).
"""
import os

import pytest
import torch

import difashion_amd as da
from oracle import unet_ref
from tests.gpu_util import DEV, rel_err
from tests.helpers import GLUE_CFG

pytestmark = pytest.mark.gpu
TOL = 3e-2


def hip_unet(cfg, params, max_batch=4):
    m = da.UNet2DConditionModel(sample_size=cfg.sample_size, in_channels=cfg.in_channels,
                                block_out_channels=cfg.block_out_channels, cross_attention_dim=cfg.cross_attention_dim,
                                attention_head_dim=cfg.num_heads, use_linear_projection=cfg.use_linear_projection,
                                max_batch=max_batch, init_seed=None)
    m.load_state_dict(params)
    return m.to(DEV).eval()


def inputs(cfg, B, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cfg.in_channels, cfg.sample_size, cfg.sample_size, generator=g)
    e = torch.randn(B, 77, cfg.cross_attention_dim, generator=g)
    return x, e


def full_size_reference(name, params, cfg, x, t, e, rows=None, want_taps=False):
    """fp32-oracle noise prediction of a full-size leg + a function ``tap_err(key, got)``.  Taken from the committed cache
    (tests/golden/unet_full_<name>.npz, written by tests/golden/make_golden_unet_full.py from this very oracle call and tied to the
    regenerated weights / inputs by a fingerprint) when it matches; otherwise -- no file, other seeds, DFH_LIVE_ORACLE=1 -- the oracle runs
    live on the host as it did before round 6.  Taps are compared on the cache's fixed 65 536-element subsample either way."""
    from tests import oracle_cache
    rec = oracle_cache.load(name, params, x, t, e)
    if rec is not None:
        print(f"[{name}] fp32 oracle from the committed cache")
        return torch.from_numpy(rec["out"]), (lambda k, got: oracle_cache.tap_rel_err(got, rec, k))
    print(f"[{name}] fp32 oracle LIVE on the host")
    taps = {}
    with torch.no_grad():
        if rows is None:
            ref = unet_ref.unet_forward(params, cfg, x, t, e, taps=taps if want_taps else None)
        else:
            ref = unet_ref.unet_forward(params, cfg, x[rows], t[rows], e[rows])
    live = {f"tap_{k}": oracle_cache.subsample(v) for k, v in taps.items()}
    return ref, (lambda k, got: oracle_cache.tap_rel_err(got, live, k))


@pytest.mark.parametrize("name,cfg", [("tiny", unet_ref.TINY), ("glue", GLUE_CFG),
                                      ("tiny_linear_proj", unet_ref.UNetConfig(sample_size=16, block_out_channels=(64, 128, 256, 256),
                                                                               cross_attention_dim=64, num_heads=(2, 2, 4, 4),
                                                                               use_linear_projection=True))])
def test_unet_matches_oracle_small(name, cfg):
    params = unet_ref.init_params(cfg, seed=3, w_std=0.05, affine_jitter=0.1)
    m = hip_unet(cfg, params)
    x, e = inputs(cfg, 3, 11)
    t = torch.tensor([7, 500, 981])
    taps = {}
    from difashion_amd import _lib
    with torch.no_grad():
        ref = unet_ref.unet_forward(params, cfg, x, t, e, taps=taps)
        _lib.census_reset()
        out = m(x.to(DEV), t.to(DEV), e.to(DEV)).sample
    cen = _lib.census()
    # the LayerNorms of the transformer blocks are folded into the projections around them wherever the producing GEMM can leave
    # row statistics (csrc/lnfold.hip); what is left runs the LayerNorm kernel
    print(name, "ln_folded", cen["ln_folded"], "layernorm launches", cen["layernorm"])
    assert cen["ln_folded"] > 0
    report = {}
    for k in ("conv_in", "down0", "down1", "down2", "down3", "mid", "up0", "up1", "up2", "up3"):
        report[k] = rel_err(m.debug_tap(k).cpu(), taps[k])
    report["out"] = rel_err(out.cpu(), ref)
    print(name, {k: f"{v:.2e}" for k, v in report.items()})
    assert out.shape == ref.shape and out.dtype == torch.float32
    assert all(v <= TOL for v in report.values()), report


def test_unet_wide_deep_levels_take_winograd_and_phase_convs():
    """A small U-Net whose two deep levels are wide (512 channels at 8x8 and 4x4): their resnet convs run as Winograd F(2x2, 3x3)
    (csrc/winograd.hip: 16 transform-domain GEMMs in one batched launch, also with the 1x1 shortcut as its own GEMM and the concatenated
    skip input), the upsamplers as four phase planes over the source image (csrc/gemm.h phase2x) -- the census shows both ran -- within
    the same tolerance."""
    from difashion_amd import _lib
    cfg = unet_ref.UNetConfig(sample_size=32, block_out_channels=(64, 64, 512, 512), cross_attention_dim=64, num_heads=(2, 2, 8, 8))
    params = unet_ref.init_params(cfg, seed=5, w_std=0.03, affine_jitter=0.1)
    m = hip_unet(cfg, params)
    x, e = inputs(cfg, 3, 13)
    t = torch.tensor([7, 500, 981])
    with torch.no_grad():
        ref = unet_ref.unet_forward(params, cfg, x, t, e)
        m(x.to(DEV), t.to(DEV), e.to(DEV))
        _lib.census_reset()
        out = m(x.to(DEV), t.to(DEV), e.to(DEV)).sample
    cen = _lib.census()
    err = rel_err(out.cpu(), ref)
    print("wide2", f"{err:.2e}", {k: v for k, v in cen.items() if v})
    assert cen["conv_wino"] == 22 and cen["conv_phase"] == 3, cen      # every resnet of the two wide levels but the 64 -> 512 one
    assert err <= TOL


def test_timestep_forms_and_return_dict():
    cfg = unet_ref.TINY
    params = unet_ref.init_params(cfg, seed=4)
    m = hip_unet(cfg, params)
    x, e = inputs(cfg, 2, 12)
    x, e = x.to(DEV), e.to(DEV)
    with torch.no_grad():
        a = m(x, torch.tensor(481, device=DEV), e, return_dict=False)[0]      # 0-d tensor (difashion.py:520)
        b = m(x, 481, encoder_hidden_states=e).sample                          # python int, kwarg form (:521)
        c = m(x, torch.tensor([481, 481]), e).sample                           # (B,) int64 on CPU (:251)
    assert torch.equal(a, b) and torch.equal(a, c)
    with torch.no_grad():      # runs are deterministic (no float atomics anywhere on the path)
        assert torch.equal(a, m(x, 481, e).sample)


def test_run_cache_reproduces_the_uncached_forward_bit_for_bit():
    """prepare_run (dfh_unet_run_cache / dfh_unet_forward_cached): the per-run constants of a sampling loop -- cross-attention K / V^T
    of every block, the schedule's time-embedding rows (difashion.py:340-357, :456) -- computed once.  The cached step must equal the
    plain forward bit for bit (same kernels on the same operands), use the cache (census), and fall back to the plain path when the
    text-state tensor, the timestep or the weights are not the prepared ones."""
    from difashion_amd import _lib
    cfg = unet_ref.TINY
    m = hip_unet(cfg, unet_ref.init_params(cfg, seed=8), max_batch=4)
    x, e = inputs(cfg, 4, 17)
    x, e = x.to(DEV), e.to(DEV)
    ts = [981, 961, 961, 501, 21]              # PLMS lists carry a duplicate entry
    with torch.no_grad():
        plain = {t: m(x, t, e).sample for t in set(ts)}
        m.prepare_run(e, ts)
        _lib.census_reset()
        for t in ts:
            assert torch.equal(m(x, t, e).sample, plain[t])
        assert _lib.census()["text_cached"] == len(ts)
        _lib.census_reset()
        assert torch.equal(m(x, 777, e).sample, m(x, torch.tensor([777.0] * 4, device=DEV), e).sample)      # not in the schedule
        assert torch.equal(m(x, 981, e.clone()).sample, plain[981])                                           # another tensor object
        assert torch.equal(m(x[:2], 981, e[:2]).sample, m(x[:2], 981, e[:2].clone()).sample)                  # another batch
        assert _lib.census()["text_cached"] == 0
        m.conv_out.bias.add_(1.0)                                                                             # weights changed: cache dropped
        torch.testing.assert_close(m(x, 981, e).sample, plain[981] + 1.0, rtol=0, atol=1e-5)
        assert _lib.census()["text_cached"] == 0
        m.end_run()


@pytest.mark.parametrize("name,cfg", [("tiny", unet_ref.TINY), ("glue", GLUE_CFG)])
@pytest.mark.parametrize("B,n,fp8", [(8, 2, False), (4, 2, False), (6, 1, False), (8, 2, True), (4, 2, True)])
def test_dup_tail_prefix_equals_the_full_batch_walk(name, cfg, B, n, fp8):
    """dfh_unet_set_dup_tail: the prompt-only branch of classifier-free guidance (difashion.py:388-427, 494-512 -- category_prompts vs
    null_prompts over the same latent / mutual / history input).  With the last n images repeating the sample of the n before them and only
    the text states differing, conv_in, the first resnet and the first block up to its self-attention run on B - n images; the result must
    equal the plain walk over all B images (to bf16 tile-choice noise -- the shorter launches may pick other tiles), the hint must be
    consumed by ONE call, and a batch whose repeated images are NOT equal must not be touched by a stale hint."""
    from difashion_amd import _lib
    m = hip_unet(cfg, unet_ref.init_params(cfg, seed=5, w_std=0.05, affine_jitter=0.1) if fp8 else unet_ref.init_params(cfg, seed=5), max_batch=8)
    if fp8:
        m.enable_fp8()           # the e4m3 walk: the prefix also carries the e4m3 attention output and its per-image maxima
    x, e = inputs(cfg, B, 23)
    x, e = x.to(DEV), e.to(DEV)
    x[B - n:] = x[B - 2 * n:B - n]
    with torch.no_grad():
        ref = m(x, 501, e).sample
        _lib.census_reset()
        m._dup_tail_once = n
        got = m(x, 501, e).sample
        used = _lib.census().get("dup_prefix", 0)
        again = m(x, 501, e).sample                        # the hint was one-shot: this is the plain walk again
    assert used == (1 if cfg.down_attn[0] else 0), used      # the census counts the transformer block that ended the shared prefix
    assert torch.equal(again, ref)
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= (6e-2 if fp8 else 2e-2) * scale, ((got - ref).abs().max().item(), scale)
    assert (got - ref).float().norm().item() <= (1.5e-2 if fp8 else 4e-3) * ref.float().norm().item()
    # the repeated images went through different text states: their outputs must differ from the images they repeat
    assert not torch.equal(got[B - n:], got[B - 2 * n:B - n])


def test_dup_tail_hint_that_does_not_hold_is_caught_under_check_mode():
    """DFH_CHECK_DUP=1 (a debugging aid for callers of dfh_unet_set_dup_tail): a hint whose repeated images are NOT equal fails loudly
    instead of producing the repeated images' outputs for the wrong inputs.  Runs in a child process (the switch is read once)."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import torch
        from oracle import unet_ref
        from tests.test_gpu_unet import hip_unet, inputs, DEV
        from difashion_amd import _lib
        cfg = unet_ref.TINY
        m = hip_unet(cfg, unet_ref.init_params(cfg, seed=5), max_batch=4)
        x, e = inputs(cfg, 4, 23); x, e = x.to(DEV), e.to(DEV)
        with torch.no_grad():
            m._dup_tail_once = 2
            try:
                m(x, 501, e); print("NOT CAUGHT")
            except _lib.DfhError as err:
                print("CAUGHT", err)
            x[2:] = x[:2]
            m._dup_tail_once = 2
            m(x, 501, e); print("EQUAL INPUTS PASS")
    """)
    env = dict(os.environ, DFH_CHECK_DUP="1")
    from tests.gpu_util import release_cached_gpu_memory
    release_cached_gpu_memory()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=600)
    assert "CAUGHT" in r.stdout and "do NOT repeat" in r.stdout and "EQUAL INPUTS PASS" in r.stdout and "NOT CAUGHT" not in r.stdout, r.stdout + r.stderr


def test_prepare_run_with_a_batch_above_max_batch_grows_the_context():
    """Two outfits x 4 items x 3 guidance branches = 24 rows against max_batch = 16 (advisor finding, round 3): prepare_run must grow the
    context the way forward() does instead of failing in dfh_unet_run_cache, and the cached steps must equal the plain forward."""
    from difashion_amd import _lib
    cfg = unet_ref.TINY
    m = hip_unet(cfg, unet_ref.init_params(cfg, seed=8), max_batch=4)
    x, e = inputs(cfg, 6, 19)
    x, e = x.to(DEV), e.to(DEV)
    with torch.no_grad():
        m.prepare_run(e, [981, 501])                  # 6 rows > max_batch = 4
        assert m.max_batch == 6
        _lib.census_reset()
        a = m(x, 981, e).sample
        assert _lib.census()["text_cached"] == 1
        m.end_run()
        assert torch.equal(a, m(x, 981, e).sample)


def test_batch_rows_independent_and_bf16_inputs():
    cfg = unet_ref.TINY
    m = hip_unet(cfg, unet_ref.init_params(cfg, seed=5))
    x, e = inputs(cfg, 3, 13)
    x, e = x.to(DEV), e.to(DEV)
    t = torch.tensor([5, 300, 900], device=DEV)
    with torch.no_grad():
        full = m(x, t, e).sample
        one = m(x[1:2], t[1:2], e[1:2]).sample
        xb = m(x.bfloat16(), t, e.bfloat16()).sample
        # independence, the rigorous form: other rows' inputs must not reach this row AT ALL -- same batch size (same kernels, tiles
        # and summation orders), rows 0 and 2 replaced: row 1 must come out bit for bit the same
        x2, e2 = x.clone(), e.clone()
        x2[0], x2[2], e2[0], e2[2] = x[2] * 3.0 + 1.0, -x[0], e[2] * 2.0, -e[0]
        other = m(x2, torch.tensor([900, 300, 5], device=DEV), e2).sample
    assert torch.equal(other[1], full[1])
    # across batch sizes the launch heuristics pick other tiles / split factors / GroupNorm kernels: two valid bf16 evaluations of
    # the same function, each ~1.6e-2 from the fp32 oracle (scripts/lnfold_diag.py prints both), differ by about as much
    assert rel_err(full[1:2], one) < 3e-2
    assert xb.dtype == torch.bfloat16 and rel_err(xb.float(), full) < 3e-2


def test_conv_in_replacement_like_the_reference():
    """difashion.py:82-93: widen conv_in 4 -> 8 channels on a constructed model, zero new channels."""
    cfg = unet_ref.TINY
    m = da.UNet2DConditionModel(sample_size=16, in_channels=4, block_out_channels=cfg.block_out_channels,
                                cross_attention_dim=64, attention_head_dim=cfg.num_heads, max_batch=2, init_seed=1)
    m.register_to_config(in_channels=8)
    with torch.no_grad():
        new = torch.nn.Conv2d(8, m.conv_in.out_channels, m.conv_in.kernel_size, m.conv_in.stride, m.conv_in.padding)
        new.weight.zero_()
        new.weight[:, :4].copy_(m.conv_in.weight)
        new.bias.copy_(m.conv_in.bias)
        old = m.conv_in
        m.conv_in = new
    m = m.to(DEV).eval()
    x, e = inputs(cfg, 2, 14)
    with torch.no_grad():
        y8 = m(x.to(DEV), 10, e.to(DEV)).sample
        x2 = x.clone()
        x2[:, 4:] = 123.0          # zero-weighted history channels must not matter
        assert torch.equal(y8, m(x2.to(DEV), 10, e.to(DEV)).sample)
    assert m.config.in_channels == 8 and y8.shape == (2, 4, 16, 16)


def test_weight_update_triggers_repack_and_checkpoint_roundtrip(tmp_path):
    cfg = unet_ref.TINY
    m = hip_unet(cfg, unet_ref.init_params(cfg, seed=6))
    x, e = inputs(cfg, 1, 15)
    x, e = x.to(DEV), e.to(DEV)
    with torch.no_grad():
        a = m(x, 3, e).sample
        m.conv_out.bias.add_(1.0)                     # in-place optimizer-style update
        b = m(x, 3, e).sample
    torch.testing.assert_close(b, a + 1.0, rtol=0, atol=1e-5)
    m.save_pretrained(str(tmp_path / "unet"))
    m2 = da.UNet2DConditionModel.from_pretrained(str(tmp_path), subfolder="unet").to(DEV).eval()
    with torch.no_grad():
        assert torch.equal(m2(x, 3, e).sample, b)


def test_cpu_tensors_fail_loudly():
    cfg = unet_ref.TINY
    m = da.UNet2DConditionModel(sample_size=16, in_channels=8, block_out_channels=cfg.block_out_channels,
                                cross_attention_dim=64, attention_head_dim=cfg.num_heads, init_seed=None)
    x, e = inputs(cfg, 1, 16)
    with pytest.raises(da.DfhError, match="no CPU fallback"):
        m(x, 1, e)


@pytest.mark.timeout(900)
def test_unet_sd15_full_size_matches_oracle():
    """SD-1.5 shape (859.5 M parameters, 64x64x8 input), B=1, weights N(0, 0.02) seed 0."""
    cfg = unet_ref.SD15
    params = unet_ref.init_params(cfg, seed=0)
    x, e = inputs(cfg, 1, 123)
    t = torch.tensor([481])
    ref, tap_err = full_size_reference("sd15_b1", params, cfg, x, t, e, want_taps=True)
    m = hip_unet(cfg, params, max_batch=1)
    del params
    with torch.no_grad():
        out = m(x.to(DEV), t.to(DEV), e.to(DEV)).sample
    report = {k: tap_err(k, m.debug_tap(k).cpu()) for k in ("conv_in", "down0", "down2", "mid", "up1", "up3")}
    report["out"] = rel_err(out.cpu(), ref)
    print("sd15", {k: f"{v:.2e}" for k, v in report.items()})
    assert all(v <= TOL for v in report.values()), report


@pytest.mark.timeout(2400)
def test_unet_sd15_full_size_batch16_matches_oracle():
    """The workload bench.py times (BASELINE configs[1]): ONE SD-1.5 forward at batch 16 = 4 items x 4 CFG branches
    (DiFashion/models/difashion.py:456-577: x_in rows are branch-major, every row carries its own timestep here), weights seed 0.
    At this batch the launch heuristics take the kernels the bench runs -- the 256 x 160 wide tile, the producer-statistics
    GroupNorm, split-K at the 16x16 / 8x8 levels, the 32x32x16 attention -- which the B=1 test never reaches; the census
    (dfh_census_*) asserts that they actually ran.  Tolerance as stated at the top: <= 3e-2 on the output and the block taps;
    the e4m3 linears (BASELINE configs[4]) <= 6e-2 on the same inputs."""
    from difashion_amd import _lib
    cfg = unet_ref.SD15
    params = unet_ref.init_params(cfg, seed=0)
    x, e = inputs(cfg, 16, 123)
    t = torch.tensor([981, 981, 981, 981, 741, 741, 741, 741, 501, 501, 501, 501, 21, 21, 21, 21])
    ref, tap_err = full_size_reference("sd15_b16", params, cfg, x, t, e, want_taps=True)
    m = hip_unet(cfg, params, max_batch=16)
    del params
    xd, td, ed = x.to(DEV), t.to(DEV), e.to(DEV)
    with torch.no_grad():
        m(xd, td, ed)                       # first call packs; the census below sees one clean forward
        torch.cuda.synchronize()
        _lib.census_reset()
        out = m(xd, td, ed).sample
        torch.cuda.synchronize()
    cen = _lib.census()
    report = {k: tap_err(k, m.debug_tap(k).cpu()) for k in ("conv_in", "down0", "down1", "down2", "mid", "up1", "up2", "up3")}
    report["out"] = rel_err(out.cpu(), ref)
    print("sd15 B=16", {k: f"{v:.2e}" for k, v in report.items()})
    print("census", {k: v for k, v in cen.items() if v})
    assert all(v <= TOL for v in report.values()), report
    assert cen["gemm_wide"] > 0, cen                                   # chip-filling conv / GEGLU launches on the 256-row tile
    assert cen["gstat_written"] > 0 and cen["gn_pre"] > 0, cen         # GroupNorm statistics from the producing GEMM's epilogue
    assert cen["splitk_reduce"] + cen["splitk_fused"] > 0, cen         # split-K at the deep levels
    assert cen["attention_x32"] > 0, cen                               # the 32x32x16 attention kernel
    assert cen["gemm_lean"] + cen["gemm_row"] > 0, cen                 # the short-K token linears
    assert cen["ln_folded"] >= 45 and cen["layernorm"] <= 3, cen       # LayerNorm folded into its consumers (all but the 8x8 block)
    assert cen["conv_wino"] == 24 and cen["conv_phase"] == 3, cen      # Winograd at the 16x16 / 8x8 levels, phase-plane upsamplers
    assert cen["mlp_fused"] == 5, cen                                  # the five 64x64-level feed-forwards as one kernel each (mlp_fused2.hip)
    assert cen["gn_folded"] == 5, cen                                  # ... and their entry GroupNorms folded into proj_in (norm.h GnFoldArgs)
    m.enable_fp8()
    with torch.no_grad():
        m(xd, td, ed)
        torch.cuda.synchronize()
        _lib.census_reset()
        out8 = m(xd, td, ed).sample
        torch.cuda.synchronize()
    cen8 = _lib.census()
    e8 = rel_err(out8.cpu(), ref)
    print("sd15 B=16 fp8", f"{e8:.2e}", {k: v for k, v in cen8.items() if v})
    # BASELINE configs[4] as named: per transformer block proj_in, q|k, v, to_out, q (cross), to_out (cross), ff.net.0, ff.net.2 and
    # proj_out on gemm_fp8_kernel = 9 x 16 blocks; no LayerNorm-fed bf16 linear is left, the 3x3 convs stay bf16
    assert e8 <= 6e-2 and cen8["gemm_fp8"] == 9 * 16, cen8
    assert cen8["attention_fp8"] == 0, cen8
    # ... and with the self-attention products on the e4m3 MFMA as well (opt-in: slower on this model, profiles/r04): all 16 self-attention
    # launches (head dims 40 / 80 / 160) on attention_fp8_kernel, same tolerance
    m.enable_fp8(True, attention=True)
    with torch.no_grad():
        m(xd, td, ed)
        torch.cuda.synchronize()
        _lib.census_reset()
        out8a = m(xd, td, ed).sample
        torch.cuda.synchronize()
    cen8a = _lib.census()
    e8a = rel_err(out8a.cpu(), ref)
    print("sd15 B=16 fp8 + fp8 attention", f"{e8a:.2e}")
    assert e8a <= 6e-2 and cen8a["attention_fp8"] == 16 and cen8a["gemm_fp8"] == 9 * 16, cen8a
    assert cen8["gemm_lean"] + cen8["gemm_8wave"] + cen8["gemm_row"] < cen["gemm_lean"] + cen["gemm_8wave"] + cen["gemm_row"], (cen, cen8)


@pytest.mark.timeout(2400)
def test_unet_sd15_full_size_batch64_matches_oracle_rows():
    """The reference's OWN inference call (DiFashion/inf4eval.py:521-524: 4 outfits per fashion_generation call in GOR mode) makes the
    U-Net batch 4 outfits x 4 items x 4 guidance branches = 64.  One full-size SD-1.5 forward at that batch: the rows of a U-Net batch
    never interact, so the fp32 oracle is evaluated on four of the 64 rows (first, last, two inside; their own timesteps) and every one
    must hold the stated tolerance (<= 3e-2 relative L2); the census shows which tiles the launch heuristics pick at this batch (the
    deep levels finally have 4x the pixel rows: fewer split-K launches than at batch 16)."""
    from difashion_amd import _lib
    cfg = unet_ref.SD15
    params = unet_ref.init_params(cfg, seed=0)
    x, e = inputs(cfg, 64, 777)
    t = torch.tensor([981, 741, 501, 21]).repeat_interleave(16)
    rows = [0, 21, 42, 63]
    ref, _ = full_size_reference("sd15_b64_rows", params, cfg, x, t, e, rows=rows)
    m = hip_unet(cfg, params, max_batch=64)
    del params
    xd, td, ed = x.to(DEV), t.to(DEV), e.to(DEV)
    with torch.no_grad():
        m(xd, td, ed)
        torch.cuda.synchronize()
        _lib.census_reset()
        out = m(xd, td, ed).sample
        torch.cuda.synchronize()
    cen = _lib.census()
    errs = [rel_err(out[r:r + 1].cpu(), ref[i:i + 1]) for i, r in enumerate(rows)]
    print("sd15 B=64 rows", rows, [f"{v:.2e}" for v in errs])
    print("census B=64", {k: v for k, v in cen.items() if v})
    assert torch.isfinite(out).all()
    assert all(v <= TOL for v in errs), errs
    assert cen["gemm_wide"] + cen["gemm_row"] > 0 and cen["attention_x32"] > 0 and cen["ln_folded"] >= 45, cen


@pytest.mark.timeout(900)
def test_unet_sd2base_full_size_matches_oracle():
    """SD-2-base shape (865.9 M parameters: linear projections, 1024-wide text states, head dim 64 at every level), B=1."""
    cfg = unet_ref.SD2BASE
    params = unet_ref.init_params(cfg, seed=0)
    x, _ = inputs(cfg, 1, 321)
    e = torch.randn(1, 77, cfg.cross_attention_dim, generator=torch.Generator().manual_seed(322))
    t = torch.tensor([731])
    ref, _ = full_size_reference("sd2base_b1", params, cfg, x, t, e)
    m = hip_unet(cfg, params, max_batch=1)
    del params
    with torch.no_grad():
        out = m(x.to(DEV), t.to(DEV), e.to(DEV)).sample
    err = rel_err(out.cpu(), ref)
    print("sd2base", f"{err:.2e}")
    assert err <= TOL
    # fp8 leg (BASELINE configs[4]) on the linear-projection variant: head dims 64 (5 / 10 / 20 heads), cross dim 1024
    from difashion_amd import _lib
    m.enable_fp8()
    with torch.no_grad():
        m(x.to(DEV), t.to(DEV), e.to(DEV))
        torch.cuda.synchronize()
        _lib.census_reset()
        out8 = m(x.to(DEV), t.to(DEV), e.to(DEV)).sample
        torch.cuda.synchronize()
    e8 = rel_err(out8.cpu(), ref)
    print("sd2base fp8", f"{e8:.2e}")
    assert e8 <= 6e-2 and _lib.census()["gemm_fp8"] == 9 * 16


@pytest.mark.parametrize("name,cfg", [("tiny", unet_ref.TINY), ("tiny_sd2", unet_ref.UNetConfig(
    sample_size=16, block_out_channels=(64, 128, 256, 256), cross_attention_dim=64, num_heads=(1, 2, 4, 4), use_linear_projection=True))])
def test_fp8_linears_vs_oracle(name, cfg):
    """BASELINE configs[4]: the LayerNorm-fed projections (attn1 q|k / v, attn2 q, GEGLU input) in e4m3 with per-token x
    per-channel scales.  Stated tolerance vs the fp32 oracle: relative L2 <= 6e-2 on the noise prediction (3 mantissa bits on
    60 % of the linear-class operands; bf16 path: 3e-2, measured ~1.5e-2); the fp8 and bf16 walks of the SAME model must also
    stay within 5e-2 of each other, and switching back must reproduce the bf16 result bit for bit."""
    params = unet_ref.init_params(cfg, seed=9, w_std=0.05, affine_jitter=0.1)
    m = hip_unet(cfg, params)
    x, e = inputs(cfg, 3, 31)
    t = torch.tensor([981, 500, 21], device=DEV)
    with torch.no_grad():
        ref = unet_ref.unet_forward(params, cfg, x, t.cpu(), e)
        y16 = m(x.to(DEV), t, e.to(DEV)).sample
        m.enable_fp8()
        y8 = m(x.to(DEV), t, e.to(DEV)).sample
        m.enable_fp8(False)
        y16b = m(x.to(DEV), t, e.to(DEV)).sample
    e16, e8, d = rel_err(y16.cpu(), ref), rel_err(y8.cpu(), ref), rel_err(y8, y16)
    print(name, f"bf16 vs oracle {e16:.2e}, fp8 vs oracle {e8:.2e}, fp8 vs bf16 {d:.2e}")
    # fp8-vs-bf16 spread measured with all nine linears per block in e4m3 (round 4, gpurun d_tests.log): tiny 4.74e-2, tiny_sd2 4.63e-2
    # (3.1e-2 / 3.0e-2 with the round-3 set of four) -- hence 5.5e-2 here, not the 5e-2 of the four-linear docstring above
    assert e8 <= 6e-2 and d <= 5.5e-2 and d > 0.0
    assert torch.equal(y16, y16b)


def test_fp8_walk_with_the_run_cache_equals_the_plain_fp8_forward():
    """The fp8 walk under prepare_run: the cross-attention maxima of V (what the e4m3 attention output is scaled by) come from the run
    cache instead of being recomputed -- same values, so the cached step equals the plain fp8 forward bit for bit."""
    cfg = unet_ref.TINY
    m = hip_unet(cfg, unet_ref.init_params(cfg, seed=9, w_std=0.05, affine_jitter=0.1), max_batch=4)
    m.enable_fp8()
    x, e = inputs(cfg, 4, 17)
    x, e = x.to(DEV), e.to(DEV)
    with torch.no_grad():
        plain = m(x, 501, e).sample
        m.prepare_run(e, [981, 501])
        cached = m(x, 501, e).sample
        m.end_run()
    assert torch.equal(plain, cached)
