#!/usr/bin/env python3
"""VERDICT r05 item 3 ("decide configs[4]"): what would e4m3 3x3 convs cost in accuracy?  CPU experiment on the fp32 oracle, no kernel needed.

The resnet 3x3 convs (and the down / up sampler convs) of chosen levels are evaluated with BOTH operands rounded to OCP e4m3 the way an
fp8 implicit-GEMM kernel would see them -- weights per output channel (amax / 448, as gemm_fp8.hip's linears), activations per tensor
(amax / 448 of the conv's input: the most favourable static scale a GroupNorm+SiLU epilogue could use) -- products and sums in fp32.
Everything else stays fp32, so the number printed is the error the e4m3 convs ALONE add; the product's fp8 walk already sits at 4.37e-2
of its 6e-2 budget with the transformer linears in e4m3 (tests/test_gpu_unet.py, SD-1.5 batch 16), and independent errors add in quadrature.

    python scripts/fp8_conv_error_probe.py            # SD-1.5 shape, B = 1, weights seed 0 (the parity legs' weights), ~1 min on 8 cores
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from oracle import unet_ref

E4M3_MAX = 448.0


def q_e4m3(x, dim=None):
    amax = x.abs().amax(dim=dim, keepdim=True) if dim is not None else x.abs().max()
    s = torch.clamp(amax / E4M3_MAX, min=1e-30)
    return (x / s).to(torch.float8_e4m3fn).float() * s


def run(levels, params, cfg, x, t, e, per_pixel=False):
    """levels: spatial sizes (64, 32, 16, 8) whose 3x3 convs run on e4m3 operands.  per_pixel: one activation scale per pixel (amax over its
    channels) instead of one per tensor -- the finest scale an implicit-GEMM kernel can apply outside the contraction."""
    real = F.conv2d

    def conv(inp, w, b=None, stride=1, padding=0, *a, **k):
        if w.shape[-1] == 3 and inp.shape[1] >= 64 and inp.shape[-1] in levels:      # conv_in (8 channels) and conv_out stay bf16 in any plan
            return real(q_e4m3(inp, dim=1 if per_pixel else None), q_e4m3(w, dim=(1, 2, 3)), b, stride, padding, *a, **k)
        return real(inp, w, b, stride, padding, *a, **k)

    unet_ref.F.conv2d = conv
    try:
        with torch.no_grad():
            return unet_ref.unet_forward(params, cfg, x, t, e)
    finally:
        unet_ref.F.conv2d = real


def main():
    cfg = unet_ref.SD15
    params = unet_ref.init_params(cfg, seed=0)
    g = torch.Generator().manual_seed(123)
    x = torch.randn(1, cfg.in_channels, cfg.sample_size, cfg.sample_size, generator=g)
    e = torch.randn(1, 77, cfg.cross_attention_dim, generator=g)
    t = torch.tensor([481])
    with torch.no_grad():
        ref = unet_ref.unet_forward(params, cfg, x, t, e)
    rel = lambda a: float((a - ref).norm() / ref.norm())
    print("e4m3 3x3 convs at levels      rel L2 of the noise prediction vs fp32      in quadrature with the measured 4.37e-2 of the fp8 linears")
    for per_pixel in (False, True):
        print("activation scale:", "one per PIXEL (amax over its channels)" if per_pixel else "one per TENSOR")
        for levels in ((64,), (32,), (64, 32), (16, 8), (64, 32, 16, 8)):
            err = rel(run(set(levels), params, cfg, x, t, e, per_pixel))
            print(f"  {str(levels):24s}    {err:.3e}                                   {(err ** 2 + 4.37e-2 ** 2) ** 0.5:.3e}   (budget 6e-2)", flush=True)


if __name__ == "__main__":
    main()
