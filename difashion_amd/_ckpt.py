"""Checkpoint-directory helpers shared by the diffusers-layout ``from_pretrained`` mirrors (reference train.py:516-554).

diffusers 0.18.2 (the reference's pin, README.md:27) writes ``diffusion_pytorch_model.bin`` by default
(``save_pretrained(safe_serialization=False)``) -- the reference's own save hook (train.py:518-524: ``unet``, ``unet_ema``,
``fashion_encoder``, ``fashion_encoder_ema``) therefore produces ``.bin`` directories, and many published snapshots carry
only ``.bin``.  ``load_weights`` takes ``.safetensors`` when present and falls back to ``.bin`` (``torch.load(weights_only=True)``),
honouring diffusers' ``variant`` infix (``diffusion_pytorch_model.fp16.safetensors``).  A requested variant that does not
exist is an error (diffusers raises too): silently loading the plain file would hand the caller other weights than asked for.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

WEIGHTS_STEM = "diffusion_pytorch_model"


def weight_candidates(variant: Optional[str] = None, stems=(WEIGHTS_STEM,)):
    return [(f"{stem}.{variant}" if variant else stem) + ext for stem in stems for ext in (".safetensors", ".bin")]


# transformers writes ``model.safetensors`` (>= 4.3x with safe serialization) or ``pytorch_model.bin`` (4.32.1 default; what the
# text_encoder/ directory of the published SD-1.5 / SD-2 snapshots holds next to it)
TRANSFORMERS_STEMS = ("model", "pytorch_model")


def load_weights(directory: str, variant: Optional[str] = None, stems=(WEIGHTS_STEM,)) -> Dict[str, torch.Tensor]:
    tried = []
    for name in weight_candidates(variant, stems):
        f = os.path.join(directory, name)
        tried.append(name)
        if not os.path.isfile(f):
            continue
        if name.endswith(".safetensors"):
            from safetensors.torch import load_file
            return load_file(f)
        sd = torch.load(f, map_location="cpu", weights_only=True)
        return sd.get("state_dict", sd) if isinstance(sd, dict) else sd
    what = f"no '{variant}' variant weights" if variant else "no weights"
    raise FileNotFoundError(f"{what} in {directory!r}: looked for {', '.join(tried)}")


# diffusers < 0.15 named the single-head VAE mid-block attention differently; diffusers 0.18.2 renames on load
# (``_convert_deprecated_attention_blocks``).  The published SD-1.5 / SD-2-base VAE weights the reference loads at
# difashion.py:74 still use the old names.
_DEPRECATED_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def remap_deprecated_vae_attention(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        parts = k.split(".")
        if len(parts) >= 3 and "attentions" in parts and parts[-2] in _DEPRECATED_ATTN:
            parts[-2] = _DEPRECATED_ATTN[parts[-2]]
            k = ".".join(parts)
            if v.dim() == 4 and v.shape[-2:] == (1, 1):        # some exports stored the projections as 1x1 convs
                v = v[:, :, 0, 0]
        elif len(parts) >= 3 and "attentions" in parts and parts[-2] in ("to_q", "to_k", "to_v", "0") and v.dim() == 4 \
                and v.shape[-2:] == (1, 1):
            v = v[:, :, 0, 0]
        out[k] = v
    return out
