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
GRAD_TOL = 6e-2
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
