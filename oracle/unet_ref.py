"""Pure-PyTorch (CPU, fp32) restatement of the conditional U-Net DiFashion calls.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  PARITY UNPINNED vs diffusers:
the reference owns no U-Net code; it imports ``UNet2DConditionModel`` from
diffusers 0.18.2 (reference DiFashion/models/difashion.py:10, constructed
:77-79, widened conv_in :82-93, called :249-253 and :518-523).  This file
restates that published architecture from its public description
 using the same ATen primitives diffusers composes:
conv2d, group_norm, silu, linear, layer_norm, gelu (erf), softmax, nearest
interpolate.

Everything is functional: ``unet_forward(params, cfg, sample, timestep, ehs)``
where ``params`` is a dict keyed by diffusers state-dict names (Appendix A.4).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class UNetConfig:
    """Subset of the diffusers UNet2DConditionModel config the path depends on."""

    sample_size: int = 64
    in_channels: int = 8            # widened 4 -> 8 at difashion.py:83-85
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    cross_attention_dim: int = 768
    # diffusers calls this field "attention_head_dim" but it is the NUMBER of heads
    num_heads: Tuple[int, ...] = (8, 8, 8, 8)
    use_linear_projection: bool = False
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    # which down blocks carry cross-attention (SD: first three)
    down_attn: Tuple[bool, ...] = (True, True, True, False)

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    @property
    def up_attn(self) -> Tuple[bool, ...]:
        return tuple(reversed(self.down_attn))


SD15 = UNetConfig()
SD2BASE = UNetConfig(cross_attention_dim=1024, num_heads=(5, 10, 20, 20), use_linear_projection=True)
# small config used by golden fixtures and fast parity tests: same topology,
# every width a legal size for the HIP kernels (channels % 32 == 0, head dims 32/64/128)
TINY = UNetConfig(sample_size=16, block_out_channels=(64, 128, 256, 256), cross_attention_dim=64,
                  num_heads=(2, 2, 2, 2))


# --------------------------------------------------------------------------- #
# parameter table (names + shapes), the structural cross-check of Appendix A.5
# --------------------------------------------------------------------------- #
def _resnet_shapes(prefix: str, cin: int, cout: int, temb: int, out: Dict[str, Tuple[int, ...]]):
    out[f"{prefix}.norm1.weight"] = (cin,)
    out[f"{prefix}.norm1.bias"] = (cin,)
    out[f"{prefix}.conv1.weight"] = (cout, cin, 3, 3)
    out[f"{prefix}.conv1.bias"] = (cout,)
    out[f"{prefix}.time_emb_proj.weight"] = (cout, temb)
    out[f"{prefix}.time_emb_proj.bias"] = (cout,)
    out[f"{prefix}.norm2.weight"] = (cout,)
    out[f"{prefix}.norm2.bias"] = (cout,)
    out[f"{prefix}.conv2.weight"] = (cout, cout, 3, 3)
    out[f"{prefix}.conv2.bias"] = (cout,)
    if cin != cout:
        out[f"{prefix}.conv_shortcut.weight"] = (cout, cin, 1, 1)
        out[f"{prefix}.conv_shortcut.bias"] = (cout,)


def _attn_shapes(prefix: str, c: int, cross: int, linear: bool, out: Dict[str, Tuple[int, ...]]):
    out[f"{prefix}.norm.weight"] = (c,)
    out[f"{prefix}.norm.bias"] = (c,)
    pshape = (c, c) if linear else (c, c, 1, 1)
    out[f"{prefix}.proj_in.weight"] = pshape
    out[f"{prefix}.proj_in.bias"] = (c,)
    tb = f"{prefix}.transformer_blocks.0"
    for n in ("norm1", "norm2", "norm3"):
        out[f"{tb}.{n}.weight"] = (c,)
        out[f"{tb}.{n}.bias"] = (c,)
    for a, kv in (("attn1", c), ("attn2", cross)):
        out[f"{tb}.{a}.to_q.weight"] = (c, c)
        out[f"{tb}.{a}.to_k.weight"] = (c, kv)
        out[f"{tb}.{a}.to_v.weight"] = (c, kv)
        out[f"{tb}.{a}.to_out.0.weight"] = (c, c)
        out[f"{tb}.{a}.to_out.0.bias"] = (c,)
    out[f"{tb}.ff.net.0.proj.weight"] = (8 * c, c)
    out[f"{tb}.ff.net.0.proj.bias"] = (8 * c,)
    out[f"{tb}.ff.net.2.weight"] = (c, 4 * c)
    out[f"{tb}.ff.net.2.bias"] = (c,)
    out[f"{prefix}.proj_out.weight"] = pshape
    out[f"{prefix}.proj_out.bias"] = (c,)


def up_block_channels(cfg: UNetConfig) -> List[List[Tuple[int, int, int]]]:
    """Per up block, per resnet: (hidden_in, skip_in, out) channel counts (Appendix A.2)."""
    boc = cfg.block_out_channels
    rev = tuple(reversed(boc))
    res = []
    out_ch = rev[0]
    for i in range(len(boc)):
        prev = out_ch
        out_ch = rev[i]
        in_ch = rev[min(i + 1, len(boc) - 1)]
        blk = []
        for j in range(cfg.layers_per_block + 1):
            skip = in_ch if j == cfg.layers_per_block else out_ch
            hid = prev if j == 0 else out_ch
            blk.append((hid, skip, out_ch))
        res.append(blk)
    return res


def param_shapes(cfg: UNetConfig) -> Dict[str, Tuple[int, ...]]:
    """Ordered {diffusers key: shape} for the whole U-Net (Appendix A.4)."""
    s: Dict[str, Tuple[int, ...]] = {}
    boc = cfg.block_out_channels
    temb = cfg.time_embed_dim
    s["conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3)
    s["conv_in.bias"] = (boc[0],)
    s["time_embedding.linear_1.weight"] = (temb, boc[0])
    s["time_embedding.linear_1.bias"] = (temb,)
    s["time_embedding.linear_2.weight"] = (temb, temb)
    s["time_embedding.linear_2.bias"] = (temb,)
    ch = boc[0]
    for i, oc in enumerate(boc):
        for j in range(cfg.layers_per_block):
            _resnet_shapes(f"down_blocks.{i}.resnets.{j}", ch if j == 0 else oc, oc, temb, s)
            if cfg.down_attn[i]:
                _attn_shapes(f"down_blocks.{i}.attentions.{j}", oc, cfg.cross_attention_dim,
                             cfg.use_linear_projection, s)
        if i != len(boc) - 1:
            s[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (oc, oc, 3, 3)
            s[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (oc,)
        ch = oc
    mid = boc[-1]
    _resnet_shapes("mid_block.resnets.0", mid, mid, temb, s)
    _attn_shapes("mid_block.attentions.0", mid, cfg.cross_attention_dim, cfg.use_linear_projection, s)
    _resnet_shapes("mid_block.resnets.1", mid, mid, temb, s)
    for i, blk in enumerate(up_block_channels(cfg)):
        for j, (hid, skip, oc) in enumerate(blk):
            _resnet_shapes(f"up_blocks.{i}.resnets.{j}", hid + skip, oc, temb, s)
            if cfg.up_attn[i]:
                _attn_shapes(f"up_blocks.{i}.attentions.{j}", oc, cfg.cross_attention_dim,
                             cfg.use_linear_projection, s)
        if i != len(boc) - 1:
            oc = blk[0][2]
            s[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (oc, oc, 3, 3)
            s[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (oc,)
    s["conv_norm_out.weight"] = (boc[0],)
    s["conv_norm_out.bias"] = (boc[0],)
    s["conv_out.weight"] = (cfg.out_channels, boc[0], 3, 3)
    s["conv_out.bias"] = (cfg.out_channels,)
    return s


def param_count(cfg: UNetConfig) -> int:
    return sum(math.prod(v) for v in param_shapes(cfg).values())


def init_params(cfg: UNetConfig, seed: int = 0, w_std: float = 0.02, affine_jitter: float = 0.0,
                dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Synthetic weights: W ~ N(0, w_std) from a CPU generator; biases 0 and
    norm gamma=1 / beta=0 unless ``affine_jitter`` > 0, in which case biases, gamma and beta are
    perturbed by N(0, affine_jitter) so parity tests exercise every bias / affine path."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    p: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        is_norm = ".norm" in name or name.startswith("conv_norm_out")
        if name.endswith(".weight") and not is_norm:
            t = torch.randn(shape, generator=g, dtype=torch.float32) * w_std
        elif name.endswith(".weight"):
            t = torch.ones(shape)
            if affine_jitter:
                t = t + torch.randn(shape, generator=g) * affine_jitter
        else:
            t = torch.zeros(shape)
            if affine_jitter:
                t = t + torch.randn(shape, generator=g) * affine_jitter
        p[name] = t.to(dtype)
    return p


# --------------------------------------------------------------------------- #
# forward
# --------------------------------------------------------------------------- #
def timestep_embedding(timesteps: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def _resnet(p, pre, x, emb, groups, eps):
    h = F.group_norm(x, groups, p[f"{pre}.norm1.weight"], p[f"{pre}.norm1.bias"], eps)
    h = F.silu(h)
    h = F.conv2d(h, p[f"{pre}.conv1.weight"], p[f"{pre}.conv1.bias"], padding=1)
    t = F.linear(F.silu(emb), p[f"{pre}.time_emb_proj.weight"], p[f"{pre}.time_emb_proj.bias"])
    h = h + t[:, :, None, None]
    h = F.group_norm(h, groups, p[f"{pre}.norm2.weight"], p[f"{pre}.norm2.bias"], eps)
    h = F.silu(h)
    h = F.conv2d(h, p[f"{pre}.conv2.weight"], p[f"{pre}.conv2.bias"], padding=1)
    if f"{pre}.conv_shortcut.weight" in p:
        x = F.conv2d(x, p[f"{pre}.conv_shortcut.weight"], p[f"{pre}.conv_shortcut.bias"])
    return x + h


def attention(q_in, kv_in, wq, wk, wv, wo, bo, heads):
    """softmax(QK^T / sqrt(d)) V with bias-free q/k/v projections and a biased out projection."""
    B, N, C = q_in.shape
    q = F.linear(q_in, wq)
    k = F.linear(kv_in, wk)
    v = F.linear(kv_in, wv)
    d = C // heads
    q = q.view(B, N, heads, d).transpose(1, 2)
    k = k.view(B, -1, heads, d).transpose(1, 2)
    v = v.view(B, -1, heads, d).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    o = torch.matmul(torch.softmax(s, dim=-1), v)
    o = o.transpose(1, 2).reshape(B, N, C)
    return F.linear(o, wo, bo)


def _transformer(p, pre, x, ehs, heads, groups, linear):
    B, C, H, W = x.shape
    res = x
    h = F.group_norm(x, groups, p[f"{pre}.norm.weight"], p[f"{pre}.norm.bias"], 1e-6)
    if not linear:
        h = F.conv2d(h, p[f"{pre}.proj_in.weight"], p[f"{pre}.proj_in.bias"])
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    else:
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = F.linear(h, p[f"{pre}.proj_in.weight"], p[f"{pre}.proj_in.bias"])
    tb = f"{pre}.transformer_blocks.0"
    n = F.layer_norm(h, (C,), p[f"{tb}.norm1.weight"], p[f"{tb}.norm1.bias"], 1e-5)
    h = h + attention(n, n, p[f"{tb}.attn1.to_q.weight"], p[f"{tb}.attn1.to_k.weight"],
                      p[f"{tb}.attn1.to_v.weight"], p[f"{tb}.attn1.to_out.0.weight"],
                      p[f"{tb}.attn1.to_out.0.bias"], heads)
    n = F.layer_norm(h, (C,), p[f"{tb}.norm2.weight"], p[f"{tb}.norm2.bias"], 1e-5)
    h = h + attention(n, ehs, p[f"{tb}.attn2.to_q.weight"], p[f"{tb}.attn2.to_k.weight"],
                      p[f"{tb}.attn2.to_v.weight"], p[f"{tb}.attn2.to_out.0.weight"],
                      p[f"{tb}.attn2.to_out.0.bias"], heads)
    n = F.layer_norm(h, (C,), p[f"{tb}.norm3.weight"], p[f"{tb}.norm3.bias"], 1e-5)
    ff = F.linear(n, p[f"{tb}.ff.net.0.proj.weight"], p[f"{tb}.ff.net.0.proj.bias"])
    a, gate = ff.chunk(2, dim=-1)
    ff = a * F.gelu(gate)
    h = h + F.linear(ff, p[f"{tb}.ff.net.2.weight"], p[f"{tb}.ff.net.2.bias"])
    if not linear:
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        h = F.conv2d(h, p[f"{pre}.proj_out.weight"], p[f"{pre}.proj_out.bias"])
    else:
        h = F.linear(h, p[f"{pre}.proj_out.weight"], p[f"{pre}.proj_out.bias"])
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return h + res


def time_embed(p, cfg: UNetConfig, timestep, batch: int, dtype=torch.float32):
    t = timestep
    if not torch.is_tensor(t):
        t = torch.tensor([t], dtype=torch.int64)
    if t.dim() == 0:
        t = t[None]
    t = t.expand(batch)
    te = timestep_embedding(t, cfg.block_out_channels[0]).to(dtype)
    e = F.linear(te, p["time_embedding.linear_1.weight"], p["time_embedding.linear_1.bias"])
    return F.linear(F.silu(e), p["time_embedding.linear_2.weight"], p["time_embedding.linear_2.bias"])


def unet_forward(p: Dict[str, torch.Tensor], cfg: UNetConfig, sample: torch.Tensor, timestep,
                 encoder_hidden_states: torch.Tensor, taps: Dict[str, torch.Tensor] | None = None):
    """UNet2DConditionModel.forward(sample, timestep, encoder_hidden_states).sample (Appendix A.2).

    ``taps`` (optional dict) receives named intermediate activations for layer-level parity tests.
    """
    G, eps = cfg.norm_num_groups, cfg.norm_eps
    lin = cfg.use_linear_projection
    ehs = encoder_hidden_states
    emb = time_embed(p, cfg, timestep, sample.shape[0], sample.dtype)
    h = F.conv2d(sample, p["conv_in.weight"], p["conv_in.bias"], padding=1)
    if taps is not None:
        taps["emb"] = emb
        taps["conv_in"] = h
    skips = [h]
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            h = _resnet(p, f"down_blocks.{i}.resnets.{j}", h, emb, G, eps)
            if cfg.down_attn[i]:
                h = _transformer(p, f"down_blocks.{i}.attentions.{j}", h, ehs, cfg.num_heads[i], G, lin)
            skips.append(h)
        if i != nb - 1:
            h = F.conv2d(h, p[f"down_blocks.{i}.downsamplers.0.conv.weight"],
                         p[f"down_blocks.{i}.downsamplers.0.conv.bias"], stride=2, padding=1)
            skips.append(h)
        if taps is not None:
            taps[f"down{i}"] = h
    h = _resnet(p, "mid_block.resnets.0", h, emb, G, eps)
    h = _transformer(p, "mid_block.attentions.0", h, ehs, cfg.num_heads[-1], G, lin)
    h = _resnet(p, "mid_block.resnets.1", h, emb, G, eps)
    if taps is not None:
        taps["mid"] = h
    rev_heads = tuple(reversed(cfg.num_heads))
    for i in range(nb):
        for j in range(cfg.layers_per_block + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = _resnet(p, f"up_blocks.{i}.resnets.{j}", h, emb, G, eps)
            if cfg.up_attn[i]:
                h = _transformer(p, f"up_blocks.{i}.attentions.{j}", h, ehs, rev_heads[i], G, lin)
        if i != nb - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, p[f"up_blocks.{i}.upsamplers.0.conv.weight"],
                         p[f"up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
        if taps is not None:
            taps[f"up{i}"] = h
    h = F.group_norm(h, G, p["conv_norm_out.weight"], p["conv_norm_out.bias"], eps)
    h = F.silu(h)
    return F.conv2d(h, p["conv_out.weight"], p["conv_out.bias"], padding=1)


class OracleUNet(torch.nn.Module):
    """nn.Module wrapper with the call surface difashion.py uses on ``self.unet``
    (difashion.py:84-93 conv_in access, :99 config.sample_size, :249-253 / :518-523 calls)."""

    class _Out:
        def __init__(self, sample):
            self.sample = sample

    class _Cfg(dict):
        __getattr__ = dict.__getitem__

    def __init__(self, cfg: UNetConfig, params: Dict[str, torch.Tensor]):
        super().__init__()
        self.cfg = cfg
        self.config = self._Cfg(sample_size=cfg.sample_size, in_channels=cfg.in_channels)
        self._names = list(params.keys())
        self._p = torch.nn.ParameterDict({k.replace(".", "__"): torch.nn.Parameter(v.clone()) for k, v in params.items()})

    def params(self):
        return {k: self._p[k.replace(".", "__")] for k in self._names}

    def forward(self, sample, timestep, encoder_hidden_states, return_dict: bool = True):
        out = unet_forward(self.params(), self.cfg, sample, timestep, encoder_hidden_states)
        return self._Out(out) if return_dict else (out,)
