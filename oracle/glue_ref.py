"""CPU restatement of the reference-owned glue around the U-Net.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  PINNED: every function here is
checked against golden vectors captured from the real
/root/reference/DiFashion/models/difashion.py (tests/golden/make_golden.py ->
tests/golden/*.npz; tests/test_oracle_glue.py).

Follows, by function:
  mutual_encoder      difashion.py:21-46   (MutualEncoder.forward, eval mode)
  mutual_mean         difashion.py:160-170 (training: MEAN of the 3 siblings)
  mutual_sum          difashion.py:475-489 (sampling: SUM of the 3 siblings; generated
                                            siblings noisy, given siblings clean)
  train_forward       difashion.py:147-267 (DiFashion.forward after the VAE encode)
  compute_snr         difashion.py:635-657
  cfg_plan            difashion.py:309-325,388-427,494-512 (which branches are stacked)
  cfg_combine         difashion.py:525-566
  sample_outfits      difashion.py:330-577 (fashion_generation denoising loop)

VAE / CLIP / tokenizer are out of scope: callers pass the
tensors those models would have produced (latents, null_latent, prompt embeddings).
History lookup policy (difashion.py:177-184,379-386) is the caller's: the
already-selected ``hist_latents`` rows are an explicit input.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ MutualEncoder
def mutual_encoder(p: Dict[str, torch.Tensor], x: torch.Tensor,
                   dropout_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """tanh(W2 . drop(leaky_relu(W1 . flat(x) + b1)) + b2); keys ``mlp.0.*`` / ``mlp.3.*``.

    ``dropout_mask`` (N, hid) of {0, 1/(1-p)} reproduces nn.Dropout(0.1) in train mode; None = eval.
    """
    n = x.shape[0]
    h = F.linear(x.reshape(n, -1), p["mlp.0.weight"], p["mlp.0.bias"])
    h = F.leaky_relu(h, 0.01)
    if dropout_mask is not None:
        h = h * dropout_mask
    y = torch.tanh(F.linear(h, p["mlp.3.weight"], p["mlp.3.bias"]))
    return y.reshape(x.shape)


def mutual_mean(noisy: torch.Tensor, olen: int) -> torch.Tensor:
    """difashion.py:162-169: per outfit, row j = sum_k w[j,k]*x_k with w = (1-eye)/(olen-1).

    The reference accumulates with Python ``sum`` starting from int 0 in sibling order, including
    the zero-weight own slot; the same order is kept here so fp32 rounding matches."""
    out = []
    w = torch.ones(olen, olen).masked_fill(torch.eye(olen) > 0, 0.0)
    w = w / torch.sum(w, dim=1)
    for idx in range(0, noisy.shape[0], olen):
        grp = noisy[idx:idx + olen]
        for row in w:
            acc = 0
            for wk, e in zip(row, grp):
                acc = acc + wk * e
            out.append(acc)
    return torch.stack(out)


def mutual_sum(olists: torch.Tensor, all_latents: torch.Tensor, prev_latents: torch.Tensor) -> torch.Tensor:
    """difashion.py:439-450,475-489: for every blank slot, sum of the outfit's other slots where
    given items contribute their clean latent and blank ones the current generated latent."""
    bsz, olen = olists.shape
    gen = (olists == 0)
    out = []
    # running index of generated items, outfit-major (difashion.py:441-449)
    gidx = torch.cumsum(gen.reshape(-1).long(), 0).reshape(bsz, olen) - 1
    for o in range(bsz):
        for i in range(olen):
            if not gen[o, i]:
                continue
            acc = 0
            for k in range(olen):
                wk = 0.0 if k == i else 1.0
                src = prev_latents[gidx[o, k]] if gen[o, k] else all_latents[o * olen + k]
                acc = acc + wk * src
            out.append(acc)
    return torch.stack(out)


# ------------------------------------------------------------------ training loss
def compute_snr(alphas_cumprod: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
    a = (alphas_cumprod ** 0.5)[timesteps].float()
    s = ((1.0 - alphas_cumprod) ** 0.5)[timesteps].float()
    return (a / s) ** 2


def train_forward(unet: Callable, enc_params: Dict[str, torch.Tensor], sched, *,
                  latents: torch.Tensor, noise: torch.Tensor, timesteps_outfit: torch.Tensor,
                  null_latent: torch.Tensor, hist_latents: torch.Tensor,
                  ehs: torch.Tensor, null_prompt: torch.Tensor,
                  random_p: Optional[torch.Tensor], random_p_cate: Optional[torch.Tensor],
                  olen: int = 4, eta: float = 0.1, mask_ratio: Optional[float] = 0.2,
                  coupling_mask_ratio: float = 0.3, cate_mask_ratio: Optional[float] = 0.2,
                  snr_gamma: Optional[float] = None, use_history: bool = True,
                  use_mutual_guidance: bool = True, dropout_mask: Optional[torch.Tensor] = None,
                  taps: Optional[dict] = None) -> torch.Tensor:
    """DiFashion.forward from the point the latents exist (difashion.py:147-267).

    ``timesteps_outfit`` is the (bsz,) draw of :154; ``random_p`` / ``random_p_cate`` are the two
    torch.rand draws of :188 / :236; ``hist_latents`` the rows chosen at :177-184 (before masking).
    """
    n = latents.shape[0]
    t = timesteps_outfit.repeat_interleave(olen).long()
    noisy = sched.add_noise(latents, noise, t)
    if use_mutual_guidance:
        mutual = mutual_encoder(enc_params, mutual_mean(noisy, olen), dropout_mask)
    else:
        mutual = torch.stack([null_latent] * n)
    hist = hist_latents.clone()
    masked_mutual = mutual.clone()
    if mask_ratio is not None:
        if use_history and use_mutual_guidance:
            image_mask = random_p < mask_ratio + coupling_mask_ratio
            hist[image_mask] = null_latent
            mutual_mask = (random_p >= mask_ratio) & (random_p < 2 * mask_ratio + coupling_mask_ratio)
            masked_mutual[mutual_mask] = null_latent
        elif use_history:
            hist[random_p < mask_ratio] = null_latent
        elif use_mutual_guidance:
            masked_mutual[random_p < mask_ratio] = null_latent
    x = (1 - eta) * noisy + eta * masked_mutual
    x = torch.cat([x, hist], dim=1)
    ehs = ehs.clone()
    if cate_mask_ratio is not None:
        cate_mask = random_p_cate < cate_mask_ratio
        ehs[cate_mask] = null_prompt[0]
    if sched.config.prediction_type == "epsilon":
        target = noise
    elif sched.config.prediction_type == "v_prediction":
        target = sched.get_velocity(latents, noise, t)
    else:
        raise ValueError(f"Unknown prediction type {sched.config.prediction_type}")
    pred = unet(x, t, ehs)
    if taps is not None:
        taps.update(x_in=x, timesteps=t, ehs=ehs, target=target, pred=pred)
    if snr_gamma is None:
        return F.mse_loss(pred.float(), target.float(), reduction="mean")
    snr = compute_snr(sched.alphas_cumprod, t)
    w = torch.stack([snr, snr_gamma * torch.ones_like(t)], dim=1).min(dim=1)[0] / snr
    loss = F.mse_loss(pred.float(), target.float(), reduction="none")
    return (loss.mean(dim=list(range(1, loss.dim()))) * w).mean()


# ------------------------------------------------------------------ CFG sampler
def cfg_plan(cate_scale: float, hist_scale: float, mutual_scale: float,
             use_history: bool = True, use_mutual_guidance: bool = True) -> Tuple[str, int]:
    """Which guidance mode fashion_generation selects (difashion.py:309-325) and how many
    replicas of the batch it stacks."""
    h = use_history and hist_scale > 1.0
    m = use_mutual_guidance and mutual_scale > 1.0
    c = cate_scale > 1.0
    if h and m and c:
        return "full", 4
    if c:
        if h:
            return "cate_hist", 3
        if m:
            return "cate_mutual", 3
        return "cate", 2
    if h and m:
        return "hist_mutual", 2     # hist stack [hist,null] AND mutual stack [mutual,null]; combined as "hist"
    if h:
        return "hist", 2
    if m:
        return "mutual", 2
    return "none", 1


# per mode: for each stacked replica, (hist is real?, mutual is real?, prompt is real?)
_BRANCHES = {
    "full":        [(1, 1, 1), (0, 1, 1), (0, 0, 1), (0, 0, 0)],
    "cate_hist":   [(1, 1, 1), (0, 1, 1), (0, 1, 0)],
    "cate_mutual": [(1, 1, 1), (1, 0, 1), (1, 0, 0)],
    "cate":        [(1, 1, 1), (1, 1, 0)],
    "hist_mutual": [(1, 1, 1), (0, 0, 1)],
    "hist":        [(1, 1, 1), (0, 1, 1)],
    "mutual":      [(1, 1, 1), (1, 0, 1)],
    "none":        [(1, 1, 1)],
}


def cfg_combine(mode: str, eps: torch.Tensor, cate_scale: float, hist_scale: float, mutual_scale: float):
    """difashion.py:525-566."""
    if mode == "full":
        a, cm, c, u = eps.chunk(4)
        return u + hist_scale * (a - cm) + mutual_scale * (cm - c) + cate_scale * (c - u)
    if mode == "cate_hist":
        ch, c, u = eps.chunk(3)
        return u + hist_scale * (ch - c) + cate_scale * (c - u)
    if mode == "cate_mutual":
        cm, c, u = eps.chunk(3)
        return u + mutual_scale * (cm - c) + cate_scale * (c - u)
    if mode == "cate":
        c, u = eps.chunk(2)
        return u + cate_scale * (c - u)
    if mode in ("hist", "hist_mutual"):
        h, u = eps.chunk(2)
        return u + hist_scale * (h - u)
    if mode == "mutual":
        m, u = eps.chunk(2)
        return u + mutual_scale * (m - u)
    return eps


def sample_outfits(unet: Callable, enc_params: Dict[str, torch.Tensor], sched, *,
                   olists: torch.Tensor, all_latents: torch.Tensor, init_latents: torch.Tensor,
                   hist_latents: torch.Tensor, null_latent: torch.Tensor,
                   category_prompts: torch.Tensor, null_prompt: torch.Tensor,
                   num_inference_steps: int = 50, cate_scale: float = 12.0, hist_scale: float = 4.0,
                   mutual_scale: float = 5.0, eta: float = 0.1, ddim_eta: float = 0.0,
                   use_history: bool = True, use_mutual_guidance: bool = True,
                   generator=None, taps: Optional[dict] = None) -> torch.Tensor:
    """The denoising loop of fashion_generation (difashion.py:356-577) on explicit tensors.

    ``unet(x, t, ehs)`` returns the noise prediction tensor.  ``taps``, if given, records the
    U-Net inputs and combined epsilon per step index (for golden comparison).
    """
    import inspect

    mode, rep = cfg_plan(cate_scale, hist_scale, mutual_scale, use_history, use_mutual_guidance)
    br = _BRANCHES[mode]
    F_ = init_latents.shape[0]
    null_rows = torch.stack([null_latent] * F_)
    null_prompts = torch.cat([null_prompt] * F_, dim=0)
    hist_stack = torch.cat([hist_latents if b[0] else null_rows for b in br], dim=0)
    ehs = torch.cat([category_prompts if b[2] else null_prompts for b in br], dim=0)
    sched.set_timesteps(num_inference_steps)
    kwargs = {}
    sig = set(inspect.signature(sched.step).parameters.keys())
    if "eta" in sig:
        kwargs["eta"] = ddim_eta
    if "generator" in sig:
        kwargs["generator"] = generator
    latents = init_latents.clone()
    prev = latents.clone()
    for i, t in enumerate(sched.timesteps):
        x = sched.scale_model_input(torch.cat([latents] * rep), t)
        if use_mutual_guidance:
            mutual = mutual_encoder(enc_params, mutual_sum(olists, all_latents, prev))
        else:
            mutual = null_rows
        mutual_stack = torch.cat([mutual if b[1] else null_rows for b in br], dim=0)
        x = (1 - eta) * x + eta * mutual_stack
        x = torch.cat([x, hist_stack], dim=1)
        eps_all = unet(x, t, ehs)
        eps = cfg_combine(mode, eps_all, cate_scale, hist_scale, mutual_scale)
        if taps is not None:
            taps[f"x_in_{i}"] = x
            taps[f"eps_{i}"] = eps
            taps[f"unet_out_{i}"] = eps_all
            taps[f"t_{i}"] = t
            taps["ehs"] = ehs
        latents = sched.step(eps, t, latents, **kwargs, return_dict=False)[0]
        prev = latents
    return latents
