"""AutoencoderKL on the MI355X HIP path, behind the diffusers call signature.

What the reference glue requires of ``self.vae`` (DiFashion/models/difashion.py):
  * ``vae.encode(images).latent_dist.mode()`` / ``.sample()`` (:129, :144, :376, :435-437) -- images (B, 3, H, W) in [-1, 1];
  * ``vae.decode(latents / vae.config.scaling_factor, return_dict=False)[0]`` (:580);
  * ``vae.config.scaling_factor`` / ``.latent_channels`` / ``.block_out_channels`` (:75, :98, :130, :360);
  * ``vae.requires_grad_(False)`` (:106) -- the VAE is frozen: there is no backward here;
  * ``from_pretrained(path, subfolder="vae")`` with diffusers key names (:66-70).

All arithmetic runs in libdifashion_hip.so (``dfh_vae_*``, csrc/vae.hip); fp32 ``nn.Parameter``s are the master weights,
packed to bf16 GEMM layouts when they change.  No PyTorch / CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import _lib
from .unet import FrozenDict, _Node


class DiagonalGaussianDistribution:
    """diffusers' latent_dist: parameters = [mean | logvar] along the channel axis, logvar clamped to [-30, 20]."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean


class AutoencoderKLOutput:
    def __init__(self, latent_dist: DiagonalGaussianDistribution):
        self.latent_dist = latent_dist


class DecoderOutput:
    def __init__(self, sample: torch.Tensor):
        self.sample = sample


class AutoencoderKL(nn.Module):
    config_name = "config.json"
    weights_name = "diffusion_pytorch_model.safetensors"

    def __init__(self, in_channels: int = 3, out_channels: int = 3, latent_channels: int = 4,
                 block_out_channels: Sequence[int] = (128, 256, 512, 512), layers_per_block: int = 2, norm_num_groups: int = 32,
                 scaling_factor: float = 0.18215, sample_size: int = 512, init_seed: Optional[int] = 0, init_std: float = 0.02,
                 **unused):
        super().__init__()
        if len(block_out_channels) > _lib.DFH_MAX_BLOCKS:
            raise ValueError(f"at most {_lib.DFH_MAX_BLOCKS} blocks supported")
        self.config = FrozenDict(in_channels=in_channels, out_channels=out_channels, latent_channels=latent_channels,
                                 block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                 norm_num_groups=norm_num_groups, scaling_factor=scaling_factor, sample_size=sample_size)
        self._ctx = None
        self._names = None
        self._dev_buffers = None
        self._ws_bytes = 0
        self._packed_sig = None
        ctx = self._make_ctx()
        try:
            table = self._table(ctx)
        finally:
            _lib.raw().dfh_vae_destroy(ctx)
        g = torch.Generator(device="cpu")
        if init_seed is not None:
            g.manual_seed(init_seed)
        for name, shape in table:
            is_norm = "norm" in name.split(".")[-2]
            if name.endswith(".weight") and not is_norm:
                t = torch.randn(shape, generator=g) * init_std if init_seed is not None else torch.zeros(shape)
            elif name.endswith(".weight"):
                t = torch.ones(shape)
            else:
                t = torch.zeros(shape)
            parts = name.split(".")
            m = self
            for p in parts[:-1]:
                if p not in m._modules:
                    m.add_module(p, _Node())
                m = m._modules[p]
            m.register_parameter(parts[-1], nn.Parameter(t))

    # ------------------------------------------------------------------ plumbing
    def register_to_config(self, **kwargs):
        self.config.update(kwargs)

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    @property
    def dtype(self) -> torch.dtype:
        return next(self.parameters()).dtype

    def _make_ctx(self):
        cfg = self.config
        c = _lib.VAEConfigC()
        c.in_channels, c.out_channels, c.latent_channels = cfg["in_channels"], cfg["out_channels"], cfg["latent_channels"]
        c.num_blocks = len(cfg["block_out_channels"])
        for i, v in enumerate(cfg["block_out_channels"]):
            c.block_out_channels[i] = v
        c.layers_per_block, c.norm_num_groups = cfg["layers_per_block"], cfg["norm_num_groups"]
        h = C.c_void_p()
        _lib.call("dfh_vae_create", C.byref(c), C.byref(h))
        return h

    @staticmethod
    def _table(ctx):
        lib = _lib.raw()
        return [(lib.dfh_vae_param_name(ctx, i).decode(),
                 tuple(lib.dfh_vae_param_dim(ctx, i, d) for d in range(lib.dfh_vae_param_ndim(ctx, i))))
                for i in range(lib.dfh_vae_num_params(ctx))]

    def param_table(self):
        ctx = self._make_ctx()
        try:
            return self._table(ctx)
        finally:
            _lib.raw().dfh_vae_destroy(ctx)

    def __del__(self):
        try:
            if self._ctx is not None:
                _lib.raw().dfh_vae_destroy(self._ctx)
        except Exception:
            pass

    def _ensure(self, encode: bool, batch: int, size: int):
        dev = self.device
        if dev.type != "cuda":
            raise _lib.DfhError("AutoencoderKL runs only on the MI355X HIP path: move it to 'cuda' (no CPU fallback)")
        if self.dtype != torch.float32:
            raise _lib.DfhError("master parameters must stay fp32 (the kernels pack their own bf16 copies)")
        lib = _lib.raw()
        if self._ctx is None:
            self._ctx = self._make_ctx()
            self._names = [n for n, _ in self._table(self._ctx)]
            self._packed_sig = None
        need = lib.dfh_vae_workspace_bytes(self._ctx, 1 if encode else 0, batch, size)
        if self._dev_buffers is None or self._dev_buffers[0].device != dev or need > self._ws_bytes:
            a16 = self._dev_buffers[0] if self._dev_buffers is not None and self._dev_buffers[0].device == dev else \
                torch.zeros(lib.dfh_vae_arena16_bytes(self._ctx), dtype=torch.uint8, device=dev)
            a32 = self._dev_buffers[1] if self._dev_buffers is not None and self._dev_buffers[1].device == dev else \
                torch.zeros(lib.dfh_vae_arena32_bytes(self._ctx), dtype=torch.uint8, device=dev)
            repack = self._dev_buffers is None or self._dev_buffers[0].device != dev
            self._dev_buffers = None
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            _lib.call("dfh_vae_bind", self._ctx, _lib.ptr(a16), _lib.ptr(a32), _lib.ptr(ws), need)
            self._dev_buffers, self._ws_bytes = (a16, a32, ws), need
            if repack:
                self._packed_sig = None
        self.pack()

    def pack(self, force: bool = False):
        named = dict(self.named_parameters())
        plist = [named[n] for n in self._names]
        sig = tuple((p.data_ptr(), p._version) for p in plist) + (_lib.weight_epoch(),)
        if not force and sig == self._packed_sig:
            return
        arr = (C.c_void_p * len(plist))(*[p.data_ptr() for p in plist])
        _lib.call("dfh_vae_pack", self._ctx, arr, len(plist), _lib.stream_ptr())
        self._packed_sig = sig

    # ------------------------------------------------------------------ the two calls of the reference
    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        if x.dim() != 4 or x.shape[1] != self.config["in_channels"] or x.shape[2] != x.shape[3]:
            raise ValueError("images must be (B, in_channels, S, S)")
        B, _, S, _ = x.shape
        f = 2 ** (len(self.config["block_out_channels"]) - 1)
        if S % f:
            raise ValueError(f"image size must be a multiple of {f}")
        self._ensure(True, B, S)
        xin = x.to(torch.float32).contiguous()
        moments = torch.empty((B, 2 * self.config["latent_channels"], S // f, S // f), dtype=torch.float32, device=x.device)
        _lib.call("dfh_vae_encode", self._ctx, _lib.ptr(xin), _lib.ptr(moments), B, S, _lib.stream_ptr())
        dist = DiagonalGaussianDistribution(moments.to(x.dtype) if x.dtype != torch.float32 else moments)
        return AutoencoderKLOutput(dist) if return_dict else (dist,)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True, **unused):
        if z.dim() != 4 or z.shape[1] != self.config["latent_channels"] or z.shape[2] != z.shape[3]:
            raise ValueError("latents must be (B, latent_channels, s, s)")
        B, _, s, _ = z.shape
        self._ensure(False, B, s)
        f = 2 ** (len(self.config["block_out_channels"]) - 1)
        zin = z.to(torch.float32).contiguous()
        img = torch.empty((B, 4, s * f, s * f), dtype=torch.float32, device=z.device)       # 4th plane = GEMM padding
        _lib.call("dfh_vae_decode", self._ctx, _lib.ptr(zin), _lib.ptr(img), B, s, _lib.stream_ptr())
        out = img[:, :self.config["out_channels"]]
        if z.dtype != torch.float32:
            out = out.to(z.dtype)
        return DecoderOutput(out) if return_dict else (out,)

    def forward(self, sample: torch.Tensor, sample_posterior: bool = False, generator=None):
        dist = self.encode(sample).latent_dist
        z = dist.sample(generator) if sample_posterior else dist.mode()
        return self.decode(z)

    # ------------------------------------------------------------------ checkpoints (diffusers directory layout)
    def save_pretrained(self, save_directory: str, **unused):
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        cfg = dict(self.config)
        cfg["_class_name"] = "AutoencoderKL"
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(save_directory, self.weights_name))

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, variant: Optional[str] = None, **unused):
        from ._ckpt import load_weights, remap_deprecated_vae_attention
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, cls.config_name)) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        model = cls(init_seed=None, **cfg)
        # published SD VAE weights still name the mid-block attention query / key / value / proj_attn (diffusers renames on load)
        model.load_state_dict(remap_deprecated_vae_attention(load_weights(d, variant)))
        return model
