"""-m gpu: training step of the HIP U-Net against torch autograd through the fp32 CPU
oracle on identical weights and inputs.

Stated tolerance: the HIP path keeps activations, weights AND activation gradients in bf16 (fp32 accumulation,
fp32 weight-gradient accumulators), the oracle is fp32 end to end.  Bound: relative L2 error of every parameter
gradient and of d loss / d sample <= GRAD_TOL, all gradients together <= ALL_TOL (observed values are printed).
Gradients that are zero in exact arithmetic (a bias ahead of a GroupNorm whose groups hold one channel) come out
as bf16 cancellation noise of ~1e-4 of the largest gradient norm: parameters whose reference gradient norm is
below FLOOR of the largest are compared in absolute terms against that floor.
"""
import pytest
import torch

import difashion_amd as da
from oracle import unet_ref
from tests.gpu_util import DEV, rel_err
from tests.helpers import GLUE_CFG
from tests.test_gpu_unet import hip_unet, inputs

pytestmark = pytest.mark.gpu
GRAD_TOL = 8e-2
ALL_TOL = 4e-2
FLOOR = 5e-3


def oracle_grads(cfg, params, x, t, e, dout):
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    xr = x.clone().requires_grad_(True)
    out = unet_ref.unet_forward(p, cfg, xr, t, e)
    out.backward(dout)
    return out.detach(), {k: v.grad for k, v in p.items()}, xr.grad


LINEAR_CFG = unet_ref.UNetConfig(sample_size=16, block_out_channels=(64, 128, 256, 256), cross_attention_dim=64,
                                 num_heads=(2, 2, 4, 4), use_linear_projection=True)


@pytest.mark.parametrize("name,cfg", [("tiny", unet_ref.TINY), ("glue", GLUE_CFG), ("tiny_linear_proj", LINEAR_CFG)])
def test_unet_backward_matches_oracle_autograd(name, cfg):
    params = unet_ref.init_params(cfg, seed=3, w_std=0.05, affine_jitter=0.1)
    m = hip_unet(cfg, params).train()
    x, e = inputs(cfg, 3, 11)
    t = torch.tensor([7, 500, 981])
    g = torch.Generator().manual_seed(5)
    dout = torch.randn(3, cfg.out_channels, cfg.sample_size, cfg.sample_size, generator=g)
    ref_out, ref_g, ref_dx = oracle_grads(cfg, params, x, t, e, dout)

    xd = x.to(DEV).requires_grad_(True)
    out = m(xd, t.to(DEV), e.to(DEV)).sample
    assert out.requires_grad
    out.backward(dout.to(DEV))
    torch.cuda.synchronize()
    assert rel_err(out.detach().cpu(), ref_out) <= 3e-2
    scale = max(float(v.norm()) for v in ref_g.values())
    report, worst = {}, ("", 0.0)
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        got, ref = p.grad.cpu(), ref_g[k]
        nr = float(ref.norm())
        err = float((got - ref).norm()) / max(nr, FLOOR * scale)
        report[k] = err
        if err > worst[1]:
            worst = (k, err)
    dx_err = rel_err(xd.grad.cpu(), ref_dx)
    tot = (sum(float((m.get_parameter(k).grad.cpu() - ref_g[k]).norm()) ** 2 for k in ref_g) ** 0.5
           / sum(float(ref_g[k].norm()) ** 2 for k in ref_g) ** 0.5)
    print(name, f"params={len(report)} worst={worst[0]}:{worst[1]:.2e} overall={tot:.2e} dx={dx_err:.2e}")
    bad = {k: f"{v:.2e} (|g|={float(ref_g[k].norm()) / scale:.1e} of max)" for k, v in report.items() if v > GRAD_TOL}
    assert not bad, bad
    assert dx_err <= GRAD_TOL and tot <= ALL_TOL


def test_gradients_accumulate_and_frozen_parameters_are_skipped():
    cfg = unet_ref.TINY
    m = hip_unet(cfg, unet_ref.init_params(cfg, seed=4, w_std=0.05)).train()
    frozen = m.get_parameter("mid_block.attentions.0.proj_in.weight")
    frozen.requires_grad_(False)
    x, e = inputs(cfg, 2, 12)
    x, e = x.to(DEV), e.to(DEV)
    t = torch.tensor([10, 700], device=DEV)
    dout = torch.randn(2, cfg.out_channels, cfg.sample_size, cfg.sample_size, device=DEV)
    m(x, t, e).sample.backward(dout)
    g1 = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    assert frozen.grad is None and len(g1) == len(list(m.parameters())) - 1
    m(x, t, e).sample.backward(dout)          # second micro-batch: gradients add up (train.py gradient accumulation)
    torch.cuda.synchronize()
    for k, p in m.named_parameters():
        if p.grad is not None:
            assert rel_err(p.grad, 2 * g1[k]) < 1e-3, k       # fp32 atomics: order-dependent in the last bits only
    # no_grad / eval inference still goes through the plain forward and matches the training forward
    with torch.no_grad():
        a = m(x, t, e).sample
    b = m(x, t, e).sample
    assert rel_err(a, b.detach()) < 3e-2      # GEGLU / time-MLP pre-activations pass through bf16 in training mode


def test_backward_without_forward_train_fails_loudly():
    cfg = unet_ref.TINY
    m = hip_unet(cfg, unet_ref.init_params(cfg, seed=4)).train()
    x, e = inputs(cfg, 1, 1)
    out = m(x.to(DEV), 5, e.to(DEV)).sample
    out.backward(torch.ones_like(out))
    with pytest.raises(da._lib.DfhError):
        m._native_backward(torch.ones_like(out), False)     # the tape of that forward is spent


# ----------------------------------------------------------------------------- whole training step (glue + optimizer)
import glob
import os

import numpy as np

from oracle import glue_ref, sched_ref
from tests.helpers import GOLDEN, enc_params, glue_unet_params, load

TRAIN = sorted(os.path.basename(p)[6:-4] for p in glob.glob(os.path.join(GOLDEN, "train_*.npz")))


def make_encoder(rec, train=True):
    p = enc_params(rec)
    hid, flat = p["mlp.0.weight"].shape
    enc = da.MutualEncoder(cate_num=11, cate_emb_size=8, latent_channels=4, latent_size=GLUE_CFG.sample_size, hid_dim=hid)
    enc.load_state_dict({**p, "category_embedding.weight": enc.category_embedding.weight.data}, strict=True)
    return enc.to(DEV).train(train)


def batch_kwargs(rec, dev):
    d = (lambda k: rec[k].to(dev)) if dev else (lambda k: rec[k])
    gamma = float(rec["snr_gamma"])
    mask = rec.get("dropout_mask")
    return dict(latents=d("latents"), noise=d("noise"), timesteps_outfit=rec["timesteps_outfit"], null_latent=d("null_latent"),
                hist_latents=d("hist_sel"), ehs=d("ehs"), null_prompt=d("null_prompt"), random_p=rec["random_p"],
                random_p_cate=rec["random_p_cate"], snr_gamma=None if np.isnan(gamma) else gamma,
                use_history=bool(rec["use_history"]), use_mutual_guidance=bool(rec["use_mutual"]),
                dropout_mask=(mask.to(dev) if (mask is not None and dev) else mask))


@pytest.mark.parametrize("case", TRAIN)
def test_training_step_gradients_vs_oracle_autograd(case):
    """loss.backward() through the four native autograd nodes (loss, U-Net, input assembly, MutualEncoder) vs torch
    autograd through the oracle restatement of DiFashion.forward on the reference's golden batch."""
    rec = load(f"train_{case}.npz")
    params = glue_unet_params()
    pr = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    er = {k: v.clone().requires_grad_(True) for k, v in enc_params(rec).items()}
    sched_o = sched_ref.DDIMRef(prediction_type=str(rec["pred_type"]))
    loss_ref = glue_ref.train_forward(lambda x, t, e: unet_ref.unet_forward(pr, GLUE_CFG, x, t, e), er, sched_o, **batch_kwargs(rec, None))
    loss_ref.backward()

    unet = hip_unet(GLUE_CFG, params, max_batch=32).train()
    enc = make_encoder(rec)
    sched = da.DDIMScheduler(prediction_type=str(rec["pred_type"]))
    loss = da.train_forward(unet, enc, sched, **batch_kwargs(rec, DEV))
    assert loss.requires_grad
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) <= 2e-2 * abs(float(loss_ref))

    def overall(named, ref):
        num = sum(float((p.grad.cpu() - ref[k].grad).norm()) ** 2 for k, p in named if ref[k].grad is not None)
        den = sum(float(ref[k].grad.norm()) ** 2 for k, p in named if ref[k].grad is not None)
        return (num / den) ** 0.5

    u_err = overall(list(unet.named_parameters()), pr)
    msg = f"{case}: loss {float(loss):.5f} ref {float(loss_ref):.5f} unet grads {u_err:.2e}"
    assert u_err <= ALL_TOL, msg
    if bool(rec["use_mutual"]):
        named = [(k, p) for k, p in enc.named_parameters() if k.startswith("mlp.")]
        e_err = overall(named, er)
        msg += f" encoder grads {e_err:.2e}"
        for k, p in named:
            assert rel_err(p.grad.cpu(), er[k].grad) <= 8e-2, (k, msg)
        assert e_err <= GRAD_TOL, msg
    else:
        assert all(p.grad is None for p in enc.parameters())
    print(msg)


def test_fused_adamw_and_clip_match_torch():
    torch.manual_seed(0)
    shapes = [(64, 32, 3, 3), (128,), (77, 40), (5,), (256, 64)]
    ps = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt = da.FusedAdamW(ps, lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05, max_grad_norm=1.0)
    ref = torch.optim.AdamW(qs, lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    for it in range(4):
        gs = [torch.randn(s, device=DEV) * (3.0 if it % 2 == 0 else 0.01) for s in shapes]      # clipped / not clipped
        for p, q, g in zip(ps, qs, gs):
            p.grad.copy_(g)
            q.grad = g.clone()
        opt.mark_fresh()
        norm_ref = torch.nn.utils.clip_grad_norm_(qs, 1.0)
        opt.step()
        ref.step()
        assert abs(float(opt.grad_norm()) - float(norm_ref)) <= 1e-4 * float(norm_ref)
        opt.zero_grad()
        assert all(float(p.grad.abs().max()) == 0.0 for p in ps)
        for p, q in zip(ps, qs):
            torch.testing.assert_close(p.data, q.data, rtol=2e-5, atol=2e-6)
    sd = opt.state_dict()
    assert len(sd["state"]) == len(shapes) and sd["param_groups"][0]["lr"] == 1e-2
    torch.testing.assert_close(sd["state"][0]["exp_avg"], ref.state_dict()["state"][0]["exp_avg"], rtol=1e-4, atol=1e-6)


def test_ema_model_matches_diffusers_schedule():
    torch.manual_seed(1)
    ps = [torch.nn.Parameter(torch.randn(33, 7, device=DEV)), torch.nn.Parameter(torch.randn(130, device=DEV))]
    ema = da.EMAModel(ps, decay=0.999)
    shadow = [p.detach().clone() for p in ps]
    for step in range(1, 6):
        with torch.no_grad():
            for p in ps:
                p.add_(torch.randn_like(p) * 0.1)
        ema.step(ps)
        d = 0.0 if step - 1 <= 0 else min((1 + step - 1) / (10 + step - 1), 0.999)
        for s, p in zip(shadow, ps):
            s.sub_((1 - d) * (s - p.detach()))
        for a, b in zip(ema.shadow_params, shadow):
            torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)
    opt = da.FusedAdamW(ps, lr=1e-3)       # flat layout: the one-launch path gives the same numbers
    ema2 = da.EMAModel(ps, decay=0.5)
    ema2.optimization_step = 50
    with torch.no_grad():
        before = [p.detach().clone() for p in ps]
        for p in ps:
            p.add_(1.0)
    ema2.step(ps)
    for s, b, p in zip(ema2.shadow_params, before, ps):
        torch.testing.assert_close(s, b - 0.5 * (b - p.detach()), rtol=1e-6, atol=1e-6)


def test_training_loop_reduces_the_loss_on_a_fixed_batch():
    """train.py:691-711 end to end on the HIP path: loss -> backward -> clip + AdamW -> EMA, repeated on one golden batch."""
    rec = load(f"train_{TRAIN[0]}.npz")
    unet = hip_unet(GLUE_CFG, glue_unet_params(), max_batch=32).train()
    enc = make_encoder(rec)
    sched = da.DDIMScheduler(prediction_type=str(rec["pred_type"]))
    opt = da.FusedAdamW(list(unet.parameters()) + list(enc.parameters()), lr=2e-4, weight_decay=1e-2, max_grad_norm=1.0)
    ema = da.EMAModel(unet.parameters(), decay=0.9999)
    kw = batch_kwargs(rec, DEV)
    losses = [float(da.train_step(unet, enc, sched, opt, ema_unet=ema, **kw)) for _ in range(12)]
    print("losses", [f"{v:.4f}" for v in losses])
    assert losses[-1] < 0.8 * losses[0] and all(np.isfinite(losses))
    assert float(opt.grad_norm()) > 0


def test_adamw_skips_parameters_without_a_fresh_gradient_like_torch():
    """torch.optim.AdamW (the reference's optimizer, train.py:586-593) skips parameters whose .grad is None: a parameter no
    backward reached this step (MutualEncoder.category_embedding is never used in forward, difashion.py:28,39-46), a frozen
    one, or a module whose backward did not run.  The gradient views of FusedAdamW are never None -- freshness is tracked
    instead.  Weight decay must NOT shrink the skipped parameters; an EMA over them still tracks the (unchanged) values."""
    torch.manual_seed(3)
    used = torch.nn.Parameter(torch.randn(33, 17, device=DEV))
    unused = torch.nn.Parameter(torch.randn(50, device=DEV))
    frozen = torch.nn.Parameter(torch.randn(20, device=DEV), requires_grad=False)
    late = torch.nn.Parameter(torch.randn(9, 9, device=DEV))
    ps = [used, unused, frozen, late]
    qs = [torch.nn.Parameter(p.detach().clone(), requires_grad=p.requires_grad) for p in ps]
    opt = da.FusedAdamW(ps, lr=1e-2, weight_decay=0.1, max_grad_norm=1.0)
    ref = torch.optim.AdamW(qs, lr=1e-2, weight_decay=0.1)
    ema = da.EMAModel(ps[:2], decay=0.5)
    shadow0 = [t.clone() for t in ema.shadow_params]
    for it in range(3):
        x = torch.randn(17, device=DEV)
        (used @ x).square().sum().backward()             # autograd accumulation stamps ``used``
        (qs[0] @ x).square().sum().backward()
        if it == 2:                                       # ``late`` receives a gradient only in the last step
            late.square().sum().backward()
            qs[3].square().sum().backward()
        torch.nn.utils.clip_grad_norm_([q for q in qs if q.grad is not None], 1.0)
        opt.step(ema=ema)
        ref.step()
        opt.zero_grad()
        ref.zero_grad()                                   # set_to_none=True: the default of the reference's torch
        for p, q in zip(ps, qs):
            torch.testing.assert_close(p.data, q.data, rtol=2e-5, atol=2e-6)
    assert torch.equal(unused.data, qs[1].data) and torch.equal(frozen.data, qs[2].data)       # untouched, no decay
    want = shadow0[1]
    for _ in range(3):
        want = want - (1 - ema.get_decay(_ + 1)) * (want - unused.data)
    torch.testing.assert_close(ema.shadow_params[1], want, rtol=1e-6, atol=1e-7)
    # zero_grad(set_to_none=False) is torch's "zeros, not None": every trainable parameter decays again
    opt.zero_grad(set_to_none=False)
    before = unused.data.clone()
    opt.step()
    assert float((unused.data - before).abs().max()) > 0


def test_stale_lazy_gradients_are_not_applied_when_the_backward_did_not_run():
    """zero_grad(lazy_modules=[unet]) leaves the previous step's U-Net gradients in place for the next backward to overwrite.
    If that backward never runs (an encoder-only / skipped step), step() must not re-apply them."""
    cfg = unet_ref.TINY
    m = hip_unet(cfg, unet_ref.init_params(cfg, seed=3), max_batch=2).train()
    extra = torch.nn.Parameter(torch.randn(8, device=DEV))
    opt = da.FusedAdamW(list(m.parameters()) + [extra], lr=1e-2, weight_decay=0.0)
    x, e = inputs(cfg, 2, 21)
    m(x.to(DEV), torch.tensor([3, 700], device=DEV), e.to(DEV)).sample.square().mean().backward()
    opt.step()
    opt.zero_grad(lazy_modules=(m,))
    snap = [p.data.clone() for p in m.parameters()]
    extra.square().sum().backward()                       # a step in which only ``extra`` gets a gradient
    e0 = extra.data.clone()
    opt.step()
    assert all(torch.equal(a, p.data) for a, p in zip(snap, m.parameters()))
    assert not torch.equal(e0, extra.data)


def test_adamw_with_folded_ema_equals_separate_updates():
    torch.manual_seed(2)
    shapes = [(40, 24), (130,), (7, 9)]
    ps = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt_a, opt_b = da.FusedAdamW(ps, lr=1e-2, max_grad_norm=1.0), da.FusedAdamW(qs, lr=1e-2, max_grad_norm=1.0)
    ema_a, ema_b = da.EMAModel(ps[:2], decay=0.9), da.EMAModel(qs[:2], decay=0.9)      # a prefix of the parameters
    for it in range(4):
        for p, q in zip(ps, qs):
            g = torch.randn_like(p)
            p.grad.copy_(g)
            q.grad.copy_(g)
        opt_a.mark_fresh(); opt_b.mark_fresh()
        opt_a.step(ema=ema_a)          # one launch for the covered range
        opt_b.step()
        ema_b.step(qs[:2])
        for p, q in zip(ps, qs):     # not bit-equal: the clip norm is reduced with fp32 atomics (order varies run to run)
            torch.testing.assert_close(p.data, q.data, rtol=1e-5, atol=1e-7)
        for a, b in zip(ema_a.shadow_params, ema_b.shadow_params):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    assert ema_a.optimization_step == ema_b.optimization_step == 4


def test_lazy_zero_grad_overwrites_instead_of_accumulating():
    """train_step's fast path: zero_grad(lazy_modules=[unet]) leaves stale values in the U-Net gradient views and the next
    native backward stores over them; the result must equal a backward into zero-filled gradients."""
    cfg = unet_ref.TINY
    m = hip_unet(cfg, unet_ref.init_params(cfg, seed=4, w_std=0.05)).train()
    opt = da.FusedAdamW(list(m.parameters()), lr=0.0)
    x, e = inputs(cfg, 2, 12)
    x, e = x.to(DEV), e.to(DEV)
    t = torch.tensor([10, 700], device=DEV)
    dout = torch.randn(2, cfg.out_channels, cfg.sample_size, cfg.sample_size, device=DEV)
    m(x, t, e).sample.backward(dout)
    ref = opt.flat_grad.clone()
    opt.flat_grad.fill_(123.0)                      # stale garbage
    opt.zero_grad(lazy_modules=(m,))
    assert m.grads_cleared and float(opt.flat_grad.max()) == 123.0      # nothing was memset
    m(x, t, e).sample.backward(dout)
    assert not m.grads_cleared
    live = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    want = torch.cat([ref[opt._slices[id(p)][0]:opt._slices[id(p)][0] + p.numel()] for p in m.parameters()])
    assert rel_err(live, want) < 1e-3
    m(x, t, e).sample.backward(dout)                # not cleared any more: accumulates
    live2 = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert rel_err(live2, 2 * want) < 1e-3


@pytest.mark.timeout(1500)
def test_sd15_full_size_backward_matches_oracle_autograd():
    """SD-1.5 shape (859.5 M parameters), B = 2 items: every parameter gradient and d loss / d sample of the HIP backward
    against torch autograd through the fp32 oracle on the host (this exercises the C = 1280 levels, the d = 40 / 80 / 160
    attention backward and the 160-wide weight-gradient tiles at their real sizes)."""
    cfg = unet_ref.SD15
    params = unet_ref.init_params(cfg, seed=0)
    x, e = inputs(cfg, 2, 123)
    t = torch.tensor([481, 37])
    g = torch.Generator().manual_seed(9)
    dout = torch.randn(2, cfg.out_channels, cfg.sample_size, cfg.sample_size, generator=g)
    ref_out, ref_g, ref_dx = oracle_grads(cfg, params, x, t, e, dout)
    m = hip_unet(cfg, params, max_batch=2).train()
    del params
    xd = x.to(DEV).requires_grad_(True)
    out = m(xd, t.to(DEV), e.to(DEV)).sample
    out.backward(dout.to(DEV))
    torch.cuda.synchronize()
    assert rel_err(out.detach().cpu(), ref_out) <= 3e-2
    scale = max(float(v.norm()) for v in ref_g.values())
    worst, num, den = ("", 0.0), 0.0, 0.0
    for k, p in m.named_parameters():
        got, ref = p.grad.cpu(), ref_g[k]
        d = float((got - ref).norm())
        err = d / max(float(ref.norm()), FLOOR * scale)
        num += d * d
        den += float(ref.norm()) ** 2
        if err > worst[1]:
            worst = (k, err)
        p.grad = None
        ref_g[k] = None
    tot = (num / den) ** 0.5
    dx_err = rel_err(xd.grad.cpu(), ref_dx)
    print(f"sd15 backward: worst={worst[0]}:{worst[1]:.2e} overall={tot:.2e} dx={dx_err:.2e}")
    assert worst[1] <= GRAD_TOL and tot <= ALL_TOL and dx_err <= GRAD_TOL, (worst, tot, dx_err)


@pytest.mark.timeout(1500)
def test_sd15_full_size_batch32_gradients_are_the_mean_of_its_two_halves():
    """BASELINE configs[2] at its FULL size (SD-1.5 shape, 8 outfits x 4 items = 32 rows, what bench.py --mode train times): a
    size-independent property instead of the oracle (fp32 autograd through 32 rows of the 860 M-parameter net takes hours on the host).
    The loss is a mean over rows, so the gradient of the 32-row batch equals the mean of the gradients of its two 16-row halves --
    accumulated by the native backward into the same .grad views -- for every parameter; what differs is bf16 accumulation order and
    tile choice (M = 131072 vs 65536 rows: other tiles, split factors, pixel splits of the weight-gradient GEMM): two bf16 evaluations
    of the same gradient, each ~2.6e-2 from the fp32 one (test above), measured 2.4e-2 apart overall / 3.3e-2 on the worst parameter --
    bounds = the stated training tolerances halved per parameter (<= 8e-2 vs the oracle) and as stated overall (<= 4e-2).  The outputs
    of the three forwards must agree row by row as well."""
    cfg = unet_ref.SD15
    params = unet_ref.init_params(cfg, seed=0)
    m = hip_unet(cfg, params, max_batch=32).train()
    del params
    x, e = inputs(cfg, 32, 321)
    g = torch.Generator().manual_seed(10)
    t = torch.randint(0, 1000, (32,), generator=g)
    dout = torch.randn(32, cfg.out_channels, cfg.sample_size, cfg.sample_size, generator=g)
    xd, td, ed, dd = x.to(DEV), t.to(DEV), e.to(DEV), dout.to(DEV)
    out = m(xd, td, ed).sample
    out.backward(dd / 32)
    torch.cuda.synchronize()
    full = {k: p.grad.clone() for k, p in m.named_parameters()}
    for p in m.parameters():
        p.grad = None
    halves = []
    for sl in (slice(0, 16), slice(16, 32)):
        o = m(xd[sl], td[sl], ed[sl]).sample
        o.backward(dd[sl] / 32)                      # accumulates into the same gradient views
        halves.append(o.detach())
    torch.cuda.synchronize()
    assert rel_err(torch.cat(halves), out.detach()) <= 2e-2
    scale = max(float(v.norm()) for v in full.values())
    worst, num, den = ("", 0.0), 0.0, 0.0
    for k, p in m.named_parameters():
        d = float((p.grad - full[k]).norm())
        err = d / max(float(full[k].norm()), FLOOR * scale)
        num += d * d
        den += float(full[k].norm()) ** 2
        if err > worst[1]:
            worst = (k, err)
    tot = (num / den) ** 0.5
    print(f"sd15 B=32 vs 2 x B=16: worst={worst[0]}:{worst[1]:.2e} overall={tot:.2e}")
    assert worst[1] <= 6e-2 and tot <= ALL_TOL, (worst, tot)


def test_side_stream_weight_gradients_are_bit_identical_to_the_one_stream_walk(tmp_path):
    """The backward walk runs large weight-gradient GEMMs on a second stream beside the data-gradient GEMM of the same layer
    (unet_train.hip TrainRun::wgrad / join).  Same kernels, same slices, same summation order: with the side stream forced on for EVERY
    weight-gradient launch (DFH_TRAIN_SIDE_MIN_FLOP=0) every matrix gradient must be bit-identical to the one-stream walk
    (DFH_TRAIN_SIDE=0), run after run -- a missing join (a buffer overwritten while the side stream still reads it) shows up here as a
    difference.  Vector gradients (biases, norm scales) are accumulated with float atomics in either mode: compared to 1e-5."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from tests.gpu_util import release_cached_gpu_memory
    release_cached_gpu_memory()
    got = {}
    for name, env in (("one", {"DFH_TRAIN_SIDE": "0"}), ("side", {"DFH_TRAIN_SIDE_MIN_FLOP": "0"})):
        out = tmp_path / f"{name}.pt"
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "train_side_worker.py"), str(out)], env={**os.environ, **env},
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        got[name] = torch.load(out)
    mats = 0
    for rep in (0, 1):
        for k, ref in got["one"][0].items():
            for other in (got["one"][rep][k], got["side"][rep][k]):
                if ref.ndim >= 2:
                    assert torch.equal(other, ref), (k, rep, float((other - ref).abs().max()))
                    mats += 1
                else:
                    torch.testing.assert_close(other, ref, rtol=1e-5, atol=1e-6 * float(ref.abs().max() + 1e-30), msg=k)
    assert mats > 100


@pytest.mark.parametrize("cfg", [GLUE_CFG, LINEAR_CFG], ids=["glue", "linear_proj"])
def test_pack_all_writes_the_same_arenas_as_pack_plus_pack_train(cfg):
    """dfh_unet_pack_all (one read of every master weight, written to the plain AND the transposed arena: what the training step calls
    after an optimizer update) against dfh_unet_pack + dfh_unet_pack_train: both bf16 arenas and the fp32 vector arena bit for bit."""
    import ctypes as C
    from difashion_amd import _lib
    params = unet_ref.init_params(cfg, seed=11)
    m = hip_unet(cfg, params, max_batch=2).train()
    x, e = inputs(cfg, 2, 5)
    m(x.to(DEV), torch.tensor([10, 700], device=DEV), e.to(DEV))           # binds the training arenas; packs through pack_all
    torch.cuda.synchronize()
    a16, a32 = m._buffers_dev[0], m._buffers_dev[1]
    a16t = m._train_buffers[0]
    got = (a16.clone(), a32.clone(), a16t.clone())
    named = dict(m.named_parameters())
    plist = [named[n] for n in m._names]
    arr = (C.c_void_p * len(plist))(*[p.data_ptr() for p in plist])
    for buf in (a16, a32, a16t):
        buf.zero_()
    _lib.call("dfh_unet_pack", m._ctx, arr, len(plist), _lib.stream_ptr())
    _lib.call("dfh_unet_pack_train", m._ctx, arr, len(plist), _lib.stream_ptr())
    torch.cuda.synchronize()
    for name, new, ref in zip(("arena16", "arena32", "arena16t"), got, (a16, a32, a16t)):
        assert torch.equal(new, ref), (name, int((new != ref).sum()))
    assert int((a16t != 0).sum()) > a16t.numel() // 4


def test_unpack_leaves_the_gradient_norm_and_the_optimizer_uses_it():
    """dfh_unet_grad_sumsq: the gradient un-pack at the end of the native backward also leaves sum(g^2) of everything it wrote; FusedAdamW.step
    (presummed=...) then skips the U-Net's range of the flat gradient buffer when it computes the clip norm (train.py:700).  Checked:
    the number against torch over the very gradients (also when the backward ACCUMULATES into existing ones), the optimizer's norm with and
    without it, and that train_step consumes it exactly once."""
    from difashion_amd.pipeline import train_forward
    rec = load(f"train_{TRAIN[0]}.npz")
    unet = hip_unet(GLUE_CFG, glue_unet_params(), max_batch=32).train()
    enc = make_encoder(rec)
    sched = da.DDIMScheduler(prediction_type=str(rec["pred_type"]))
    opt = da.FusedAdamW(list(unet.parameters()) + list(enc.parameters()), lr=2e-4, weight_decay=1e-2, max_grad_norm=1.0)
    kw = batch_kwargs(rec, DEV)

    def torch_sumsq(params):
        return float(sum((p.grad.double() ** 2).sum() for p in params if p.grad is not None))

    for rep in range(2):                     # second round: the backward adds onto the gradients of the first (no zero_grad in between)
        train_forward(unet, enc, sched, **kw).backward()
        torch.cuda.synchronize()
        assert unet.grad_sumsq_valid
        got, want = float(unet._grad_sumsq), torch_sumsq(unet.parameters())
        assert abs(got - want) <= 1e-5 * want, (rep, got, want)
    want_all = torch_sumsq(list(unet.parameters()) + list(enc.parameters())) ** 0.5
    state = (opt.flat_param.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone())
    opt.step(presummed=(list(unet.parameters()), unet._grad_sumsq))
    n_pre = float(opt.grad_norm())
    p_pre = opt.flat_param.clone()
    opt.flat_param.copy_(state[0]); opt.exp_avg.copy_(state[1]); opt.exp_avg_sq.copy_(state[2]); opt._step -= 1
    for p in opt._pstep:
        opt._pstep[p] -= 1
    opt.step()                               # the same update with the norm from a pass over the whole flat gradient buffer
    n_full = float(opt.grad_norm())
    assert abs(n_pre - want_all) <= 1e-5 * want_all and abs(n_full - want_all) <= 1e-5 * want_all, (n_pre, n_full, want_all)
    torch.testing.assert_close(p_pre, opt.flat_param, rtol=1e-6, atol=1e-9)
    # train_step: uses it, and invalidates it
    opt.zero_grad(lazy_modules=(unet,))
    da.train_step(unet, enc, sched, opt, **kw)
    assert not unet.grad_sumsq_valid and float(opt.grad_norm()) > 0
    # validity is checked, not trusted: an in-place edit of any gradient between backward and step (loss-scale un-scaling, a manual
    # clip) withdraws it, and the optimizer then sums the squares itself
    train_forward(unet, enc, sched, **kw).backward()
    assert unet.grad_sumsq_valid
    next(iter(unet.parameters())).grad.mul_(0.5)
    assert not unet.grad_sumsq_valid


def test_segmented_backward_equals_the_monolithic_one():
    """dfh_unet_backward_begin / _next / _finish (the pieces behind the overlapped gradient all-reduce): the ranges handed
    out are disjoint, cover every written float of the packed gradient arena exactly once, and the master gradients equal
    those of dfh_unet_backward."""
    import ctypes as C
    from difashion_amd import _lib
    cfg = unet_ref.TINY
    params = unet_ref.init_params(cfg, seed=3)
    x, e = inputs(cfg, 4, 9)
    grads = []
    for segmented in (False, True):
        m = hip_unet(cfg, params, max_batch=4).train()
        out = m(x.to(DEV), torch.tensor([7, 100, 500, 900], device=DEV), e.to(DEV)).sample
        dout = torch.ones_like(out) * 0.01
        if not segmented:
            out.backward(dout)
        else:
            plist = m.grad_views()
            arr = (C.c_void_p * len(plist))(*[p.grad.data_ptr() for p in plist])
            sp = _lib.stream_ptr()
            n = _lib.call_count("dfh_unet_backward_begin", m._ctx, _lib.ptr(dout.contiguous().float()), None, 4096, sp)
            assert n > 10
            lo, hi, seen = C.c_size_t(0), C.c_size_t(0), []
            while True:
                rc = _lib.raw().dfh_unet_backward_next(m._ctx, C.byref(lo), C.byref(hi), sp)
                assert rc >= 0, _lib.last_error()
                if rc == 0:
                    break
                seen.append((lo.value, hi.value))
            _lib.call("dfh_unet_backward_finish", m._ctx, arr, len(plist), 1, sp)
            seen.sort()
            assert len(seen) > 3 and all(a[1] <= b[0] for a, b in zip(seen, seen[1:])) and all(h > l and l % 4096 == 0 for l, h in seen)
            total = m._train_buffers[1].numel() // 4
            assert seen[-1][1] <= total
        torch.cuda.synchronize()
        grads.append({n_: p.grad.clone() for n_, p in m.named_parameters()})
    for k in grads[0]:
        torch.testing.assert_close(grads[1][k], grads[0][k], rtol=1e-5, atol=1e-7, msg=k)


def test_bf16_wire_kernels_match_the_torch_formulation():
    """dist.exchange_bf16 on the device: dfh_wire_pack / dfh_wire_shard_mean / dfh_wire_unpack against the torch ops the CPU (gloo) path
    uses -- bit for bit (the replicas of a data-parallel run must agree whichever path computed a shard): round to nearest even, zero
    padding, rank-ordered fp32 sum, one division, one rounding."""
    from difashion_amd import _lib
    sp = _lib.stream_ptr
    world, n = 4, 100003                                    # a range whose length is no multiple of anything
    per = ((n + world - 1) // world + 7) // 8 * 8
    g = torch.randn(n, device=DEV) * 3.0
    wire = torch.full((world * per,), 7.0, dtype=torch.bfloat16, device=DEV)
    _lib.call("dfh_wire_pack", _lib.ptr(g), _lib.ptr(wire), n, world * per, sp())
    assert torch.equal(wire[:n], g.to(torch.bfloat16)) and float(wire[n:].abs().max()) == 0.0
    recv = (torch.randn(world, per, device=DEV) * 2.0).to(torch.bfloat16)
    shard = torch.empty(per, dtype=torch.bfloat16, device=DEV)
    _lib.call("dfh_wire_shard_mean", _lib.ptr(recv), _lib.ptr(shard), world, per, sp())
    acc = recv[0].float()
    for r in range(1, world):
        acc += recv[r].float()
    assert torch.equal(shard, (acc / world).to(torch.bfloat16))
    back = torch.full((n,), -1.0, device=DEV)
    _lib.call("dfh_wire_unpack", _lib.ptr(wire), _lib.ptr(back), n, sp())
    assert torch.equal(back, wire[:n].float())
    # three ranks (no power of two: the division is a real division)
    recv3 = recv[:3].contiguous()
    _lib.call("dfh_wire_shard_mean", _lib.ptr(recv3), _lib.ptr(shard), 3, per, sp())
    assert torch.equal(shard, ((recv3[0].float() + recv3[1].float() + recv3[2].float()) / 3).to(torch.bfloat16))
