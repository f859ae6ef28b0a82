"""MutualEncoder (reference: DiFashion/models/difashion.py:21-46) on the HIP GEMM path.

``tanh(W2 . dropout(leaky_relu(W1 . flat(x) + b1)) + b2)`` with W1 (hid, C*S*S), W2 (C*S*S, hid).
State-dict keys match the reference (``category_embedding.weight`` -- unused in forward, kept so
checkpoints round-trip -- ``mlp.0.*``, ``mlp.3.*``).  At N = 4 items the layer is a weight-streaming
GEMV-like problem (16.8 MB of bf16 weights): both linears go through dfh_gemm (split-K for the
16384-deep first layer), activations fused in the epilogues.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Optional

import torch
import torch.nn as nn

from . import _lib

ACT_LEAKY, ACT_TANH = 2, 3
OUT_BF16, OUT_F32 = 0, 2


class _MutualStep(torch.autograd.Function):
    """Autograd node of the encoder MLP: backward runs dfh_act_bwd / dfh_gemm_wgrad / dfh_colsum / dfh_gemm and ADDS the
    parameter gradients into ``.grad`` (same convention as the U-Net node).  The input (sibling mean of the noisy
    latents, difashion.py:160-175) carries no gradient."""

    @staticmethod
    def forward(ctx, enc, x_bf16, dropout_mask, anchor):
        y, h, hd = enc._forward_native(x_bf16, dropout_mask)
        ctx.enc, ctx.kept = enc, (x_bf16, h, hd, y, dropout_mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        ctx.enc._backward_native(*ctx.kept, dy.contiguous().float())
        return None, None, None, None


class _Config(dict):
    """diffusers-style config: attribute and mapping access (the reference reads ``fashion_encoder.config`` both ways)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class MutualEncoder(nn.Module):
    config_name = "config.json"
    weights_name = "diffusion_pytorch_model.safetensors"

    def __init__(self, cate_num: int, cate_emb_size: int, latent_channels: int, latent_size: int, hid_dim: int, **unused):
        super().__init__()
        self.config = _Config(cate_num=cate_num, cate_emb_size=cate_emb_size, latent_channels=latent_channels,
                              latent_size=latent_size, hid_dim=hid_dim)
        self.category_embedding = nn.Embedding(cate_num, cate_emb_size)  # unused in forward (difashion.py:28)
        self.latent_channels = latent_channels
        self.latent_size = latent_size
        flat = latent_channels * latent_size * latent_size
        if flat % 8 or hid_dim % 8:
            raise ValueError("latent and hidden widths must be multiples of 8")
        self.mlp = nn.Sequential(nn.Linear(flat, hid_dim), nn.LeakyReLU(), nn.Dropout(0.1),
                                 nn.Linear(hid_dim, flat), nn.Tanh())
        self._packed = None
        self._sig = None
        self._anchor = None

    # ---- checkpoint directory layout of the reference (ModelMixin: train.py:516-554) ---------------------------------
    def register_to_config(self, **kwargs):
        self.config.update(kwargs)

    def save_pretrained(self, save_directory: str, **unused):
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        cfg = dict(self.config)
        cfg["_class_name"] = "MutualEncoder"
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(save_directory, self.weights_name))

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, variant: Optional[str] = None, **unused):
        from ._ckpt import load_weights
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, cls.config_name)) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        ctor = {k: cfg[k] for k in ("cate_num", "cate_emb_size", "latent_channels", "latent_size", "hid_dim")}
        model = cls(**ctor)
        model.register_to_config(**cfg)          # extra keys (e.g. EMA state written by EMAModel.save_pretrained) survive
        model.load_state_dict(load_weights(d, variant))
        return model

    def _pack(self):
        w1, b1, w2, b2 = self.mlp[0].weight, self.mlp[0].bias, self.mlp[3].weight, self.mlp[3].bias
        if w1.device.type != "cuda":
            raise _lib.DfhError("MutualEncoder runs on the HIP path only: move it to 'cuda'")
        sig = tuple((p.data_ptr(), p._version) for p in (w1, b1, w2, b2)) + (_lib.weight_epoch(),)
        if sig == self._sig:
            return self._packed
        dev = w1.device
        hid, flat = w1.shape
        p1 = torch.empty((hid, flat), dtype=torch.bfloat16, device=dev)
        p2 = torch.empty((flat, hid), dtype=torch.bfloat16, device=dev)
        s = _lib.stream_ptr()
        _lib.call("dfh_pack_matrix", _lib.ptr(w1.detach().float().contiguous()), _lib.ptr(p1), hid, flat, flat, 0, 0, 0, s)
        _lib.call("dfh_pack_matrix", _lib.ptr(w2.detach().float().contiguous()), _lib.ptr(p2), flat, hid, hid, 0, 0, 0, s)
        p2t = torch.empty((hid, flat), dtype=torch.bfloat16, device=dev)     # W2^T for the data gradient of the hidden layer
        _lib.call("dfh_pack_matrix_t", _lib.ptr(w2.detach().float().contiguous()), _lib.ptr(p2t), flat, hid, flat, 0, 0, 0, s)
        zero = torch.zeros(256, dtype=torch.uint8, device=dev)
        self._packed = (p1, p2, b1.detach().float().contiguous(), b2.detach().float().contiguous(), zero, p2t)
        self._sig = sig
        return self._packed

    def _gemm(self, x, K, W, bias, N, act, out, out_mode, zero, M):
        d = _lib.GemmDesc()
        d.a0, d.a0_c = x.data_ptr(), K
        d.W, d.ldw = W.data_ptr(), K
        d.M, d.N = M, N
        if bias is not None:
            d.bias = bias.data_ptr()
        d.act = act
        d.out, d.ld_out, d.out_mode = out.data_ptr(), N, out_mode
        d.zero_page = zero.data_ptr()
        d.force_order = -1
        need = _lib.raw().dfh_gemm_partial_floats(C.byref(d))
        part = None
        if need:
            part = torch.empty(need, dtype=torch.float32, device=out.device)
            d.partial, d.partial_floats = part.data_ptr(), need
        _lib.call("dfh_gemm", C.byref(d), _lib.stream_ptr())
        return part

    def _forward_native(self, x_bf16, dropout_mask):
        p1, p2, b1, b2, zero, _ = self._pack()
        n = x_bf16.shape[0]
        hid, flat = p1.shape
        h = torch.empty((n, hid), dtype=torch.bfloat16, device=x_bf16.device)
        keep = [self._gemm(x_bf16, flat, p1, b1, hid, ACT_LEAKY, h, OUT_BF16, zero, n)]
        hd = h
        if dropout_mask is not None:      # train-mode nn.Dropout(0.1) with a caller-supplied mask
            hd = (h.float() * dropout_mask).to(torch.bfloat16)
        y = torch.empty((n, flat), dtype=torch.float32, device=x_bf16.device)
        keep.append(self._gemm(hd, hid, p2, b2, flat, ACT_TANH, y, OUT_F32, zero, n))
        return y, h, hd

    def _backward_native(self, x_bf16, h, hd, y, dropout_mask, dy):
        """dy: fp32 (N, flat) gradient of the tanh output; parameter gradients are added into ``.grad``."""
        p1, p2, b1, b2, zero, p2t = self._pack()
        w1, bb1, w2, bb2 = self.mlp[0].weight, self.mlp[0].bias, self.mlp[3].weight, self.mlp[3].bias
        for p in (w1, bb1, w2, bb2):
            if p.requires_grad and p.grad is None:
                p.grad = torch.zeros_like(p)
        n = x_bf16.shape[0]
        hid, flat = p1.shape
        dev, s = dy.device, _lib.stream_ptr()
        dpre2 = torch.empty((n, flat), dtype=torch.bfloat16, device=dev)       # d tanh: dy * (1 - y^2)
        _lib.call("dfh_act_bwd", None, _lib.ptr(y), None, _lib.ptr(dy), _lib.ptr(dpre2), n * flat, ACT_TANH, 1.0, s)

        def wgrad(a, K, dY, N, p):
            if not p.requires_grad:
                return
            d = _lib.GemmDesc()
            d.a0, d.a0_c, d.M, d.N, d.zero_page = a.data_ptr(), K, n, N, zero.data_ptr()
            _lib.call("dfh_gemm_wgrad", C.byref(d), _lib.ptr(dY), N, _lib.ptr(p.grad), K, 1, s)     # few rows: one m-slice

        def colsum(dY, N, p):
            if p.requires_grad:
                _lib.call("dfh_colsum", _lib.ptr(dY), N, N, 1, n, _lib.ptr(p.grad), N, s)

        wgrad(hd, hid, dpre2, flat, w2)
        colsum(dpre2, flat, bb2)
        dhd = torch.empty((n, hid), dtype=torch.bfloat16, device=dev)
        keep = self._gemm(dpre2, flat, p2t, None, hid, 0, dhd, OUT_BF16, zero, n)
        if dropout_mask is not None:
            dhd = (dhd.float() * dropout_mask).to(torch.bfloat16)
        dpre1 = torch.empty((n, hid), dtype=torch.bfloat16, device=dev)       # leaky_relu: slope by the sign of the output
        _lib.call("dfh_act_bwd", _lib.ptr(h), None, _lib.ptr(dhd), None, _lib.ptr(dpre1), n * hid, ACT_LEAKY, 1.0, s)
        wgrad(x_bf16, flat, dpre1, hid, w1)
        colsum(dpre1, hid, bb1)
        return keep

    def forward_bf16(self, x_bf16: torch.Tensor, dropout_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x_bf16: (N, C*S*S) bf16 rows (the sibling-reduce kernel's output) -> fp32 (N, C, S, S)."""
        n = x_bf16.shape[0]
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.mlp.parameters()):
            if self._anchor is None or self._anchor.device != x_bf16.device:
                self._anchor = torch.zeros((), device=x_bf16.device, requires_grad=True)
            y = _MutualStep.apply(self, x_bf16, dropout_mask, self._anchor)
        else:
            y = self._forward_native(x_bf16, dropout_mask)[0]
        return y.view(n, self.latent_channels, self.latent_size, self.latent_size)

    def forward(self, mutual_emb: torch.Tensor) -> torch.Tensor:
        bsz = mutual_emb.shape[0]
        x = mutual_emb.reshape(bsz, -1).float().contiguous()
        xb = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        _lib.call("dfh_cast_f32_to_bf16", _lib.ptr(x), _lib.ptr(xb), x.numel(), _lib.stream_ptr())
        if self.training:
            raise NotImplementedError("train-mode dropout needs an explicit mask: use forward_bf16(x, dropout_mask)")
        return self.forward_bf16(xb)
