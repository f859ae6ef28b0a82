"""Pure-PyTorch (CPU, fp32) restatement of the CLIP text encoder the reference runs on the category prompts.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  **PARITY PINNED**: the reference owns no text-encoder code; it imports
``CLIPTextModel`` from transformers (DiFashion/models/difashion.py:16, built at :70-72, frozen at :107, called as
``self.text_encoder(input_ids)[0]`` at :224, :234, :340-342, :352; pin transformers 4.32.1, README.md:24).  Unlike diffusers,
``transformers`` IS installed in the build container, so ``tests/golden/make_golden_clip.py`` runs the real
``transformers.CLIPTextModel`` (eager attention, fp32) on seeded weights / token ids and commits what it returns as
``tests/golden/clip_*.npz``; ``tests/test_clip_cpu.py`` holds this restatement to those outputs (<= 2e-5 relative L2: the two
differ in summation order only) and ``tests/test_gpu_clip.py`` holds the HIP encoder to the same files.

Restated (transformers ``models/clip/modeling_clip.py``; the 4.32.1 arithmetic is the same, it only scales q before the product):
  CLIPTextEmbeddings : token_embedding[input_ids] + position_embedding[arange(T)]
  CLIPEncoderLayer   : x + out_proj(attention(layer_norm1(x)));  x + fc2(act(fc1(layer_norm2(x))))        (pre-LN)
  CLIPAttention      : q / k / v projections WITH bias, heads of hidden_size / num_heads, scores * head_dim**-0.5, the causal mask
                       (key position <= query position) and NO padding mask (the reference passes input_ids only), softmax in fp32
  CLIPMLP            : quick_gelu = x * sigmoid(1.702 x)  (SD-1.5, CLIP ViT-L/14)  or  gelu, erf form  (SD-2, OpenCLIP ViT-H/14)
  CLIPTextModel      : final_layer_norm on the last layer's output = ``[0]``; pooler_output = that tensor's row at argmax(input_ids)
                       when config.eos_token_id == 2 (the shipped SD configs; the only rule 4.32.1 has), else at the first
                       eos_token_id; ``hidden_states`` = (embeddings, layer 1, ..., layer L) before final_layer_norm

``params`` is keyed by the transformers 4.32.1 state-dict names (``text_model.embeddings.token_embedding.weight`` ...), which
is what the checkpoints the reference loads contain.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class CLIPTextConfig:
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    hidden_act: str = "quick_gelu"
    layer_norm_eps: float = 1e-5
    eos_token_id: int = 2              # the value inside the published SD-1.5 / SD-2 text_encoder/config.json (legacy argmax pooling)
    bos_token_id: int = 49406
    pad_token_id: int = 49407          # what the SD-1.5 tokenizer pads with ("<|endoftext|>"); SD-2 pads with 0 ("!")


SD15_CLIP = CLIPTextConfig()                                             # openai/clip-vit-large-patch14 text tower
SD2_CLIP = CLIPTextConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=23, num_attention_heads=16,
                          hidden_act="gelu", pad_token_id=0)             # OpenCLIP ViT-H/14 text tower, penultimate-layer export
TINY_CLIP = CLIPTextConfig(vocab_size=1000, hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4,
                           bos_token_id=998, pad_token_id=999)
TINY_CLIP_GELU = CLIPTextConfig(vocab_size=1000, hidden_size=96, intermediate_size=160, num_hidden_layers=2, num_attention_heads=3,
                                hidden_act="gelu", eos_token_id=999, bos_token_id=998, pad_token_id=0)


def param_shapes(cfg: CLIPTextConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) in transformers 4.32.1 state-dict order."""
    D, I = cfg.hidden_size, cfg.intermediate_size
    out = [("text_model.embeddings.token_embedding.weight", (cfg.vocab_size, D)),
           ("text_model.embeddings.position_embedding.weight", (cfg.max_position_embeddings, D))]
    for l in range(cfg.num_hidden_layers):
        p = f"text_model.encoder.layers.{l}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            out += [(p + f"self_attn.{n}.weight", (D, D)), (p + f"self_attn.{n}.bias", (D,))]
        out += [(p + "layer_norm1.weight", (D,)), (p + "layer_norm1.bias", (D,)),
                (p + "mlp.fc1.weight", (I, D)), (p + "mlp.fc1.bias", (I,)), (p + "mlp.fc2.weight", (D, I)), (p + "mlp.fc2.bias", (D,)),
                (p + "layer_norm2.weight", (D,)), (p + "layer_norm2.bias", (D,))]
    out += [("text_model.final_layer_norm.weight", (D,)), ("text_model.final_layer_norm.bias", (D,))]
    return out


def init_params(cfg: CLIPTextConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights (there is no network for checkpoints), drawn in table order from ONE CPU generator so that the golden
    script, the CPU tests and the GPU box regenerate the same tensors.  Scales chosen so the encoder is exercised, not idle: projection
    weights wide enough for peaked (non-uniform) causal softmaxes, jittered LayerNorm affine, non-zero biases."""
    g = torch.Generator().manual_seed(seed)
    D = cfg.hidden_size
    out = {}
    for name, shape in param_shapes(cfg):
        leaf = name.split(".")[-2]
        if leaf.startswith("layer_norm") or leaf == "final_layer_norm":
            t = torch.randn(shape, generator=g) * 0.1 + (1.0 if name.endswith("weight") else 0.0)
        elif name.endswith(".bias"):
            t = torch.randn(shape, generator=g) * 0.05
        elif "embedding" in leaf:
            t = torch.randn(shape, generator=g) * 0.05
        elif leaf in ("q_proj", "k_proj"):
            t = torch.randn(shape, generator=g) * (1.7 / D ** 0.5)
        else:
            t = torch.randn(shape, generator=g) * (0.7 / shape[1] ** 0.5)
        out[name] = t
    return out


def prompt_like_ids(cfg: CLIPTextConfig, batch: int, seq_len: int, seed: int = 0) -> torch.Tensor:
    """[bos, n words, eos, pad ...] rows like the tokenizer's output (data_utils.py:107-110); row 0 is the empty ("null") prompt
    (difashion.py:226-234); the last row is full (truncation=True)."""
    g = torch.Generator().manual_seed(seed)
    eos = cfg.vocab_size - 1 if cfg.eos_token_id == 2 else cfg.eos_token_id           # tokenizer eos = highest id (49407)
    words_hi = min(cfg.bos_token_id, eos, cfg.vocab_size - 2)
    ids = torch.full((batch, seq_len), cfg.pad_token_id, dtype=torch.long)
    for b in range(batch):
        n = 0 if b == 0 else (seq_len - 2 if b == batch - 1 else int(torch.randint(1, max(2, seq_len - 2), (1,), generator=g)))
        n = max(0, min(n, seq_len - 2))
        ids[b, 0] = cfg.bos_token_id
        ids[b, 1:1 + n] = torch.randint(1, words_hi, (n,), generator=g)
        ids[b, 1 + n] = eos
    return ids


def _act(x: torch.Tensor, kind: str) -> torch.Tensor:
    if kind == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    if kind == "gelu":
        return F.gelu(x)
    raise ValueError(kind)


@torch.no_grad()
def clip_text_forward(params: Dict[str, torch.Tensor], cfg: CLIPTextConfig, input_ids: torch.Tensor, output_hidden_states: bool = False):
    """-> (last_hidden_state [B, T, D], pooler_output [B, D], hidden_states tuple or None)."""
    P = params
    ids = input_ids.reshape(-1, input_ids.shape[-1])
    B, T = ids.shape
    if T > cfg.max_position_embeddings:
        raise ValueError(f"Sequence length must be less than max_position_embeddings (got {T} and {cfg.max_position_embeddings})")
    D, H = cfg.hidden_size, cfg.num_attention_heads
    d = D // H
    x = P["text_model.embeddings.token_embedding.weight"][ids] + P["text_model.embeddings.position_embedding.weight"][:T][None]
    causal = torch.full((T, T), float("-inf")).triu(1)
    hidden = [x]
    for l in range(cfg.num_hidden_layers):
        p = f"text_model.encoder.layers.{l}."
        h = F.layer_norm(x, (D,), P[p + "layer_norm1.weight"], P[p + "layer_norm1.bias"], cfg.layer_norm_eps)
        q = F.linear(h, P[p + "self_attn.q_proj.weight"], P[p + "self_attn.q_proj.bias"]).view(B, T, H, d).transpose(1, 2)
        k = F.linear(h, P[p + "self_attn.k_proj.weight"], P[p + "self_attn.k_proj.bias"]).view(B, T, H, d).transpose(1, 2)
        v = F.linear(h, P[p + "self_attn.v_proj.weight"], P[p + "self_attn.v_proj.bias"]).view(B, T, H, d).transpose(1, 2)
        w = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5 + causal, dim=-1, dtype=torch.float32)
        a = (w @ v).transpose(1, 2).reshape(B, T, D)
        x = x + F.linear(a, P[p + "self_attn.out_proj.weight"], P[p + "self_attn.out_proj.bias"])
        h = F.layer_norm(x, (D,), P[p + "layer_norm2.weight"], P[p + "layer_norm2.bias"], cfg.layer_norm_eps)
        h = _act(F.linear(h, P[p + "mlp.fc1.weight"], P[p + "mlp.fc1.bias"]), cfg.hidden_act)
        x = x + F.linear(h, P[p + "mlp.fc2.weight"], P[p + "mlp.fc2.bias"])
        hidden.append(x)
    last = F.layer_norm(x, (D,), P["text_model.final_layer_norm.weight"], P["text_model.final_layer_norm.bias"], cfg.layer_norm_eps)
    if cfg.eos_token_id == 2:
        pos = ids.argmax(dim=-1)
    else:
        pos = (ids == cfg.eos_token_id).int().argmax(dim=-1)
    pooled = last[torch.arange(B), pos]
    return last, pooled, (tuple(hidden) if output_hidden_states else None)
