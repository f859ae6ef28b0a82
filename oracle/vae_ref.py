"""Pure-PyTorch (CPU, fp32) restatement of the AutoencoderKL the reference calls on either side of the denoising path.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  PARITY UNPINNED vs diffusers: the reference owns no VAE code; it imports
``AutoencoderKL`` from diffusers 0.18.2 (DiFashion/models/difashion.py:10; ``vae.encode(images).latent_dist.mode()/sample()``
at :129, :144, :376, :435-437; ``vae.decode(latents / scaling_factor, return_dict=False)[0]`` at :580).  This file restates
the published Stable-Diffusion VAE architecture with the ATen primitives diffusers composes (conv2d, group_norm, silu,
softmax, nearest interpolate, F.pad):

  Encoder : conv_in 3->128; 4 x DownEncoderBlock2D (128, 256, 512, 512): 2 ResnetBlock2D (no time embedding, GroupNorm 32
            eps 1e-6, SiLU) + Downsample2D (F.pad(0,1,0,1) then conv3x3 stride 2 padding 0) except the last; mid block
            (resnet, single-head self-attention over the H*W tokens with GroupNorm 32 in front, biased q/k/v/out
            projections and a residual connection, resnet); GroupNorm + SiLU + conv_out -> 2 * latent_channels;
            quant_conv 1x1.  The result holds (mean, logvar); ``mode`` = mean, ``sample`` = mean + exp(0.5 logvar) eps
            with logvar clamped to [-30, 20].
  Decoder : post_quant_conv 1x1; conv_in 4->512; mid block; 4 x UpDecoderBlock2D (512, 512, 256, 128): 3 resnets +
            Upsample2D (nearest 2x then conv3x3) except the last; GroupNorm + SiLU + conv_out -> 3.

Functional: ``encode(params, cfg, x)`` / ``decode(params, cfg, z)`` with ``params`` keyed by diffusers state-dict names
(``encoder.down_blocks.0.resnets.0.conv1.weight``, ``decoder.mid_block.attentions.0.to_q.weight``, ``quant_conv.weight`` ...).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215
    sample_size: int = 512


SD_VAE = VAEConfig()
TINY_VAE = VAEConfig(block_out_channels=(32, 64, 64, 64), sample_size=32)


def _resnet_shapes(pre, cin, cout, out):
    out[f"{pre}.norm1.weight"] = (cin,); out[f"{pre}.norm1.bias"] = (cin,)
    out[f"{pre}.conv1.weight"] = (cout, cin, 3, 3); out[f"{pre}.conv1.bias"] = (cout,)
    out[f"{pre}.norm2.weight"] = (cout,); out[f"{pre}.norm2.bias"] = (cout,)
    out[f"{pre}.conv2.weight"] = (cout, cout, 3, 3); out[f"{pre}.conv2.bias"] = (cout,)
    if cin != cout:
        out[f"{pre}.conv_shortcut.weight"] = (cout, cin, 1, 1); out[f"{pre}.conv_shortcut.bias"] = (cout,)


def _mid_shapes(pre, c, out):
    _resnet_shapes(f"{pre}.resnets.0", c, c, out)
    a = f"{pre}.attentions.0"
    out[f"{a}.group_norm.weight"] = (c,); out[f"{a}.group_norm.bias"] = (c,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        out[f"{a}.{n}.weight"] = (c, c); out[f"{a}.{n}.bias"] = (c,)
    _resnet_shapes(f"{pre}.resnets.1", c, c, out)


def param_shapes(cfg: VAEConfig) -> Dict[str, Tuple[int, ...]]:
    """Name -> shape in the enumeration order of the native library's parameter table."""
    out: Dict[str, Tuple[int, ...]] = {}
    boc, L, nb = cfg.block_out_channels, cfg.layers_per_block, len(cfg.block_out_channels)
    out["encoder.conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3); out["encoder.conv_in.bias"] = (boc[0],)
    ch = boc[0]
    for i in range(nb):
        for j in range(L):
            _resnet_shapes(f"encoder.down_blocks.{i}.resnets.{j}", ch if j == 0 else boc[i], boc[i], out)
        ch = boc[i]
        if i != nb - 1:
            out[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (ch, ch, 3, 3)
            out[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (ch,)
    _mid_shapes("encoder.mid_block", boc[-1], out)
    out["encoder.conv_norm_out.weight"] = (boc[-1],); out["encoder.conv_norm_out.bias"] = (boc[-1],)
    out["encoder.conv_out.weight"] = (2 * cfg.latent_channels, boc[-1], 3, 3); out["encoder.conv_out.bias"] = (2 * cfg.latent_channels,)
    out["quant_conv.weight"] = (2 * cfg.latent_channels, 2 * cfg.latent_channels, 1, 1); out["quant_conv.bias"] = (2 * cfg.latent_channels,)
    out["post_quant_conv.weight"] = (cfg.latent_channels, cfg.latent_channels, 1, 1); out["post_quant_conv.bias"] = (cfg.latent_channels,)
    out["decoder.conv_in.weight"] = (boc[-1], cfg.latent_channels, 3, 3); out["decoder.conv_in.bias"] = (boc[-1],)
    _mid_shapes("decoder.mid_block", boc[-1], out)
    rev = tuple(reversed(boc))
    ch = rev[0]
    for i in range(nb):
        for j in range(L + 1):
            _resnet_shapes(f"decoder.up_blocks.{i}.resnets.{j}", ch if j == 0 else rev[i], rev[i], out)
        ch = rev[i]
        if i != nb - 1:
            out[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (ch, ch, 3, 3)
            out[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (ch,)
    out["decoder.conv_norm_out.weight"] = (boc[0],); out["decoder.conv_norm_out.bias"] = (boc[0],)
    out["decoder.conv_out.weight"] = (cfg.out_channels, boc[0], 3, 3); out["decoder.conv_out.bias"] = (cfg.out_channels,)
    return out


def param_count(cfg: VAEConfig) -> int:
    n = 0
    for s in param_shapes(cfg).values():
        k = 1
        for d in s:
            k *= d
        n += k
    return n


def init_params(cfg: VAEConfig, seed: int = 0, w_std: float = 0.02, affine_jitter: float = 0.0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, shape in param_shapes(cfg).items():
        is_norm = "norm" in name.split(".")[-2]
        if name.endswith(".weight") and not is_norm:
            p[name] = torch.randn(shape, generator=g) * w_std
        elif name.endswith(".weight"):
            p[name] = torch.ones(shape) + affine_jitter * torch.randn(shape, generator=g)
        else:
            p[name] = affine_jitter * torch.randn(shape, generator=g)
    return p


def _resnet(p, pre, x, groups):
    h = F.silu(F.group_norm(x, groups, p[f"{pre}.norm1.weight"], p[f"{pre}.norm1.bias"], eps=1e-6))
    h = F.conv2d(h, p[f"{pre}.conv1.weight"], p[f"{pre}.conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, groups, p[f"{pre}.norm2.weight"], p[f"{pre}.norm2.bias"], eps=1e-6))
    h = F.conv2d(h, p[f"{pre}.conv2.weight"], p[f"{pre}.conv2.bias"], padding=1)
    if f"{pre}.conv_shortcut.weight" in p:
        x = F.conv2d(x, p[f"{pre}.conv_shortcut.weight"], p[f"{pre}.conv_shortcut.bias"])
    return x + h


def _attention(p, pre, x, groups):
    """diffusers Attention as the VAE configures it: one head of width C, GroupNorm in front, residual connection."""
    B, C, H, W = x.shape
    h = F.group_norm(x, groups, p[f"{pre}.group_norm.weight"], p[f"{pre}.group_norm.bias"], eps=1e-6)
    t = h.view(B, C, H * W).transpose(1, 2)
    q = F.linear(t, p[f"{pre}.to_q.weight"], p[f"{pre}.to_q.bias"])
    k = F.linear(t, p[f"{pre}.to_k.weight"], p[f"{pre}.to_k.bias"])
    v = F.linear(t, p[f"{pre}.to_v.weight"], p[f"{pre}.to_v.bias"])
    a = torch.softmax(q @ k.transpose(1, 2) * (C ** -0.5), dim=-1) @ v
    o = F.linear(a, p[f"{pre}.to_out.0.weight"], p[f"{pre}.to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(B, C, H, W)


def _mid(p, pre, x, groups):
    x = _resnet(p, f"{pre}.resnets.0", x, groups)
    x = _attention(p, f"{pre}.attentions.0", x, groups)
    return _resnet(p, f"{pre}.resnets.1", x, groups)


def encode_moments(p, cfg: VAEConfig, x: torch.Tensor, taps=None) -> torch.Tensor:
    """images (B, 3, H, W) -> (B, 2*latent_channels, H/8, W/8) = [mean | logvar]  (AutoencoderKL.encode before sampling)."""
    G, nb, L = cfg.norm_num_groups, len(cfg.block_out_channels), cfg.layers_per_block
    h = F.conv2d(x, p["encoder.conv_in.weight"], p["encoder.conv_in.bias"], padding=1)
    for i in range(nb):
        for j in range(L):
            h = _resnet(p, f"encoder.down_blocks.{i}.resnets.{j}", h, G)
        if i != nb - 1:
            h = F.pad(h, (0, 1, 0, 1))
            h = F.conv2d(h, p[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"],
                         p[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"], stride=2)
        if taps is not None:
            taps[f"enc_down{i}"] = h
    h = _mid(p, "encoder.mid_block", h, G)
    if taps is not None:
        taps["enc_mid"] = h
    h = F.silu(F.group_norm(h, G, p["encoder.conv_norm_out.weight"], p["encoder.conv_norm_out.bias"], eps=1e-6))
    h = F.conv2d(h, p["encoder.conv_out.weight"], p["encoder.conv_out.bias"], padding=1)
    return F.conv2d(h, p["quant_conv.weight"], p["quant_conv.bias"])


def encode(p, cfg: VAEConfig, x: torch.Tensor, sample_noise: torch.Tensor = None) -> torch.Tensor:
    """latent_dist.mode() (sample_noise None) or latent_dist.sample() with the given standard-normal noise."""
    m = encode_moments(p, cfg, x)
    mean, logvar = m.chunk(2, dim=1)
    if sample_noise is None:
        return mean
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * sample_noise


def decode(p, cfg: VAEConfig, z: torch.Tensor, taps=None) -> torch.Tensor:
    """latents (B, latent_channels, h, w) -> images (B, 3, 8h, 8w)  (AutoencoderKL.decode(...)[0])."""
    G, nb, L = cfg.norm_num_groups, len(cfg.block_out_channels), cfg.layers_per_block
    h = F.conv2d(z, p["post_quant_conv.weight"], p["post_quant_conv.bias"])
    h = F.conv2d(h, p["decoder.conv_in.weight"], p["decoder.conv_in.bias"], padding=1)
    h = _mid(p, "decoder.mid_block", h, G)
    if taps is not None:
        taps["dec_mid"] = h
    for i in range(nb):
        for j in range(L + 1):
            h = _resnet(p, f"decoder.up_blocks.{i}.resnets.{j}", h, G)
        if i != nb - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, p[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"],
                         p[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
        if taps is not None:
            taps[f"dec_up{i}"] = h
    h = F.silu(F.group_norm(h, G, p["decoder.conv_norm_out.weight"], p["decoder.conv_norm_out.bias"], eps=1e-6))
    return F.conv2d(h, p["decoder.conv_out.weight"], p["decoder.conv_out.bias"], padding=1)
