"""``CLIPTextModel`` on the MI355X HIP path, behind the transformers call signature.

What the reference requires of ``self.text_encoder`` (DiFashion/models/difashion.py):
  * ``CLIPTextModel.from_pretrained(path, subfolder="text_encoder", revision=...)`` (:70-72) -- a transformers directory
    (``config.json`` + ``model.safetensors`` / ``pytorch_model.bin``) with ``text_model.*`` keys (transformers 4.32.1, README.md:24);
  * ``text_encoder(input_ids)[0]`` -- last_hidden_state (B, 77, D) for the category prompts of a training batch (:224), the
    empty prompt (:234, :352) and the prompts of the slots a sampling call fills (:340-342); no attention mask is passed;
  * ``text_encoder.dtype`` (:342, :411-425), ``.requires_grad_(False)`` (:107), ``.to(device)``.

All arithmetic runs in libdifashion_hip.so (``dfh_clip_encode``, csrc/clip.hip) in fp32 on the fp32 matrix instruction: the
prompts are a closed set encoded once per run (``prompts.PromptTable``), so the encoder is built to agree with the fp32 class
to summation-order noise.  The fp32 ``nn.Parameter``s are read in place (no packed copy).  No PyTorch / CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Optional

import torch
import torch.nn as nn

from . import _lib
from .unet import FrozenDict, _Node

_ACT = {"quick_gelu": 1, "gelu": 2}


class BaseModelOutputWithPooling:
    """transformers' output object as far as the reference (and ``output_hidden_states=True`` debugging) uses it: ``[0]`` /
    ``.last_hidden_state``, ``[1]`` / ``.pooler_output``, ``.hidden_states``."""

    def __init__(self, last_hidden_state, pooler_output, hidden_states=None):
        self.last_hidden_state, self.pooler_output, self.hidden_states = last_hidden_state, pooler_output, hidden_states

    def to_tuple(self):
        return tuple(v for v in (self.last_hidden_state, self.pooler_output, self.hidden_states) if v is not None)

    def __getitem__(self, i):
        return self.to_tuple()[i]

    def __iter__(self):
        return iter(self.to_tuple())

    def __len__(self):
        return len(self.to_tuple())


class CLIPTextModel(nn.Module):
    config_name = "config.json"

    def __init__(self, vocab_size: int = 49408, hidden_size: int = 768, intermediate_size: int = 3072, num_hidden_layers: int = 12,
                 num_attention_heads: int = 12, max_position_embeddings: int = 77, hidden_act: str = "quick_gelu",
                 layer_norm_eps: float = 1e-5, eos_token_id: int = 2, bos_token_id: int = 49406, pad_token_id: int = 1,
                 init_seed: Optional[int] = 0, init_std: float = 0.02, **unused):
        super().__init__()
        if hidden_act not in _ACT:
            raise ValueError(f"hidden_act {hidden_act!r}: the CLIP text towers of SD-1.5 / SD-2 use 'quick_gelu' / 'gelu'")
        self.config = FrozenDict(vocab_size=vocab_size, hidden_size=hidden_size, intermediate_size=intermediate_size,
                                 num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
                                 max_position_embeddings=max_position_embeddings, hidden_act=hidden_act, layer_norm_eps=layer_norm_eps,
                                 eos_token_id=eos_token_id, bos_token_id=bos_token_id, pad_token_id=pad_token_id)
        self._ctx = None
        self._ws = None
        ctx = self._make_ctx()
        try:
            self._names = [n for n, _ in self._table(ctx)]
            table = self._table(ctx)
        finally:
            _lib.raw().dfh_clip_destroy(ctx)
        g = torch.Generator(device="cpu")
        if init_seed is not None:
            g.manual_seed(init_seed)
        for name, shape in table:
            norm = "layer_norm" in name.split(".")[-2]
            if name.endswith(".weight") and not norm:
                t = torch.randn(shape, generator=g) * init_std if init_seed is not None else torch.zeros(shape)
            elif name.endswith(".weight"):
                t = torch.ones(shape)
            else:
                t = torch.zeros(shape)
            m = self
            parts = name.split(".")
            for p in parts[:-1]:
                if p not in m._modules:
                    m.add_module(p, _Node())
                m = m._modules[p]
            m.register_parameter(parts[-1], nn.Parameter(t))

    # ------------------------------------------------------------------ plumbing
    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    @property
    def dtype(self) -> torch.dtype:
        return next(self.parameters()).dtype

    def _make_ctx(self):
        cfg = self.config
        c = _lib.CLIPConfigC(cfg["vocab_size"], cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"],
                             cfg["num_attention_heads"], cfg["max_position_embeddings"], _ACT[cfg["hidden_act"]], cfg["layer_norm_eps"])
        h = C.c_void_p()
        _lib.call("dfh_clip_create", C.byref(c), C.byref(h))
        return h

    @staticmethod
    def _table(ctx):
        lib = _lib.raw()
        return [(lib.dfh_clip_param_name(ctx, i).decode(),
                 tuple(lib.dfh_clip_param_dim(ctx, i, d) for d in range(lib.dfh_clip_param_ndim(ctx, i))))
                for i in range(lib.dfh_clip_num_params(ctx))]

    def param_table(self):
        ctx = self._make_ctx()
        try:
            return self._table(ctx)
        finally:
            _lib.raw().dfh_clip_destroy(ctx)

    def __del__(self):
        try:
            if self._ctx is not None:
                _lib.raw().dfh_clip_destroy(self._ctx)
        except Exception:
            pass

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """Accepts the 4.32.1 layout (``text_model.*``, what published checkpoints hold), the flattened layout of newer transformers
        releases, and drops the ``position_ids`` buffer old checkpoints carry."""
        sd = {}
        for k, v in state_dict.items():
            if k.endswith("position_ids"):
                continue
            sd[k if k.startswith("text_model.") else "text_model." + k] = v
        return super().load_state_dict(sd, strict=strict, **kw)

    # ------------------------------------------------------------------ text_encoder(input_ids)
    @torch.no_grad()
    def forward(self, input_ids: Optional[torch.Tensor] = None, attention_mask=None, position_ids=None, output_attentions=None,
                output_hidden_states: Optional[bool] = None, return_dict: Optional[bool] = None):
        if input_ids is None:
            raise ValueError("You have to specify input_ids")
        if attention_mask is not None or position_ids is not None or output_attentions:
            raise NotImplementedError("the reference calls text_encoder(input_ids) only (difashion.py:224,340): no padding mask, "
                                      "default positions, no attention maps on this path")
        dev = self.device
        if dev.type != "cuda":
            raise _lib.DfhError("CLIPTextModel runs only on the MI355X HIP path: move it to 'cuda' (no CPU fallback)")
        if self.dtype != torch.float32:
            raise _lib.DfhError("parameters must stay fp32 (the kernels read them in place)")
        cfg = self.config
        shape = tuple(input_ids.shape)
        T = shape[-1]
        if T > cfg["max_position_embeddings"]:
            raise ValueError(f"Sequence length must be less than max_position_embeddings (got `sequence length`: {T} and "
                             f"max_position_embeddings: {cfg['max_position_embeddings']}")
        # nn.Embedding raises on out-of-range ids too.  Checked where the ids LIVE: tokenizer output is a CPU tensor (no device sync then);
        # ids that already sit on the GPU cost one sync, and the encoder runs once per run (PromptTable.build)
        lo, hi = int(input_ids.min()), int(input_ids.max())
        ids = input_ids.reshape(-1, T).to(device=dev, dtype=torch.int64).contiguous()
        B = ids.shape[0]
        if lo < 0 or hi >= cfg["vocab_size"]:
            raise IndexError(f"input_ids out of range [0, {cfg['vocab_size']}): min {lo}, max {hi}")
        lib = _lib.raw()
        if self._ctx is None:
            self._ctx = self._make_ctx()
        need = lib.dfh_clip_workspace_bytes(self._ctx, B, T)
        if self._ws is None or self._ws.device != dev or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        named = dict(self.named_parameters())
        plist = [named[n] for n in self._names]
        if any(p.device != dev or not p.is_contiguous() for p in plist):
            raise _lib.DfhError("all parameters must be contiguous and on one device")
        arr = (C.c_void_p * len(plist))(*[p.data_ptr() for p in plist])
        D, L = cfg["hidden_size"], cfg["num_hidden_layers"]
        last = torch.empty((B, T, D), dtype=torch.float32, device=dev)
        pooled = torch.empty((B, D), dtype=torch.float32, device=dev)
        hs = torch.empty((L + 1, B, T, D), dtype=torch.float32, device=dev) if output_hidden_states else None
        _lib.call("dfh_clip_encode", self._ctx, arr, len(plist), _lib.ptr(ids), _lib.ptr(last), _lib.ptr(pooled), int(cfg["eos_token_id"]),
                  _lib.ptr(hs), _lib.ptr(self._ws), self._ws.numel(), B, T, _lib.stream_ptr())
        out = BaseModelOutputWithPooling(last, pooled, tuple(hs[i] for i in range(L + 1)) if hs is not None else None)
        return out if return_dict is None or return_dict else out.to_tuple()

    # ------------------------------------------------------------------ checkpoints (transformers directory layout)
    def save_pretrained(self, save_directory: str, **unused):
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        cfg = dict(self.config)
        cfg.update(architectures=["CLIPTextModel"], model_type="clip_text_model")
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(save_directory, "model.safetensors"))

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, variant: Optional[str] = None, revision=None, **unused):
        from ._ckpt import TRANSFORMERS_STEMS, load_weights
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, cls.config_name)) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        model = cls(init_seed=None, **cfg)
        model.load_state_dict(load_weights(d, variant, TRANSFORMERS_STEMS))
        return model
