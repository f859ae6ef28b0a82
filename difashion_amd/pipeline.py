"""Host-side counterparts of the reference's two callers of the U-Net, on explicit tensors:

  * ``sample_outfits``  <- DiFashion.fashion_generation, denoising loop difashion.py:356-577
  * ``train_forward``   <- DiFashion.forward from the point the latents exist, difashion.py:147-267

VAE / CLIP / tokenizer are out of scope: callers pass what those models would
have produced (clean latents, null latent, prompt states) and the history rows already selected
(lookup policy is the caller's).  Every tensor op on the path is a HIP kernel from
libdifashion_hip.so: sibling reduce, MutualEncoder GEMMs, input assembly with CFG replica stacking,
the U-Net, guidance combine fused with the scheduler update.  Torch only allocates and indexes.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import torch

from . import _lib
from .mutual import MutualEncoder
from .schedulers import DDIMScheduler, PNDMScheduler, STEP_NONE

# guidance modes, replica order and which conditions are real per replica (hist, mutual, prompt):
# fashion_generation difashion.py:309-325 (mode), :388-427 and :494-512 (stacking), :525-566 (combine)
CFG_NONE, CFG_FULL, CFG_CATE_HIST, CFG_CATE_MUTUAL, CFG_CATE, CFG_HIST, CFG_MUTUAL = range(7)
_BRANCHES = {
    "full": (CFG_FULL, [(1, 1, 1), (0, 1, 1), (0, 0, 1), (0, 0, 0)]),
    "cate_hist": (CFG_CATE_HIST, [(1, 1, 1), (0, 1, 1), (0, 1, 0)]),
    "cate_mutual": (CFG_CATE_MUTUAL, [(1, 1, 1), (1, 0, 1), (1, 0, 0)]),
    "cate": (CFG_CATE, [(1, 1, 1), (1, 1, 0)]),
    "hist_mutual": (CFG_HIST, [(1, 1, 1), (0, 0, 1)]),
    "hist": (CFG_HIST, [(1, 1, 1), (0, 1, 1)]),
    "mutual": (CFG_MUTUAL, [(1, 1, 1), (1, 0, 1)]),
    "none": (CFG_NONE, [(1, 1, 1)]),
}


def guidance_plan(cate_scale: float, hist_scale: float, mutual_scale: float, use_history: bool = True,
                  use_mutual_guidance: bool = True) -> str:
    h = use_history and hist_scale > 1.0
    m = use_mutual_guidance and mutual_scale > 1.0
    c = cate_scale > 1.0
    if h and m and c:
        return "full"
    if c:
        return "cate_hist" if h else ("cate_mutual" if m else "cate")
    if h and m:
        return "hist_mutual"
    return "hist" if h else ("mutual" if m else "none")


def sampling_tables(olists: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Index/weight tables for dfh_mutual_reduce in sampling (difashion.py:439-450,475-489): for each
    blank slot (outfit-major order) and each slot k of its outfit: >= 0 -> row of the generated
    latents, < 0 -> -(row+1) of the given items' clean latents; own slot gets weight 0."""
    ol = olists.cpu()
    bsz, olen = ol.shape
    gen = ol == 0
    gidx = torch.cumsum(gen.reshape(-1).long(), 0).reshape(bsz, olen) - 1
    tab, wt = [], []
    for o in range(bsz):
        for i in range(olen):
            if not gen[o, i]:
                continue
            tab.append([int(gidx[o, k]) if gen[o, k] else -(o * olen + k + 1) for k in range(olen)])
            wt.append([0.0 if k == i else 1.0 for k in range(olen)])
    return torch.tensor(tab, dtype=torch.int32), torch.tensor(wt, dtype=torch.float32)


def training_tables(n_items: int, olen: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Training (difashion.py:162-169): row j = mean of the other olen-1 noisy siblings; weights are
    the reference's ones/sum(ones) = fp32(1/(olen-1)), own slot 0."""
    w = torch.ones(olen, olen).masked_fill(torch.eye(olen) > 0, 0.0)
    w = w / torch.sum(w, dim=1)
    tab = [[(j // olen) * olen + k for k in range(olen)] for j in range(n_items)]
    wt = [w[j % olen].tolist() for j in range(n_items)]
    return torch.tensor(tab, dtype=torch.int32), torch.tensor(wt, dtype=torch.float32)


_TABLE_CACHE = {}


def _training_tables_on(n_items: int, olen: int, dev):
    """Device-resident copy of training_tables (a pageable host->device copy every step would synchronise the host)."""
    key = (n_items, olen, str(dev))
    if key not in _TABLE_CACHE:
        tab, wt = training_tables(n_items, olen)
        _TABLE_CACHE[key] = (tab.to(dev), wt.to(dev))
    return _TABLE_CACHE[key]


def _u8(flags, dev):
    return torch.tensor(flags, dtype=torch.uint8, device=dev)


class OutfitSampler:
    """State of one CFG sampling run (the loop body of fashion_generation, difashion.py:456-577).

    ``prepare`` does the per-run setup (replica stacks, index tables, buffers); ``step(i)`` is one
    denoising step = sibling reduce + MutualEncoder + input assembly + U-Net forward at batch R*F +
    guidance combine + scheduler update.  Nothing is allocated inside ``step`` on the DDIM path."""

    def __init__(self, unet, fashion_encoder: MutualEncoder, scheduler):
        self.unet, self.enc, self.sched = unet, fashion_encoder, scheduler

    @torch.no_grad()
    def prepare(self, *, olists, all_latents, init_latents, hist_latents, null_latent, category_prompts, null_prompt,
                num_inference_steps=50, cate_scale=12.0, hist_scale=4.0, mutual_scale=5.0, eta=0.1, ddim_eta=0.0,
                use_history=True, use_mutual_guidance=True, generator=None, keep_eps=False):
        dev = init_latents.device
        if dev.type != "cuda":
            raise _lib.DfhError("sampling runs on the HIP path only (device tensors required)")
        self.mode_name = guidance_plan(cate_scale, hist_scale, mutual_scale, use_history, use_mutual_guidance)
        self.mode, br = _BRANCHES[self.mode_name]
        self.R, self.F = len(br), init_latents.shape[0]
        self.CL, S = init_latents[0].numel(), init_latents.shape[-1]
        f32 = dict(dtype=torch.float32, device=dev)
        self.latents = init_latents.to(**f32).clone().contiguous()
        self.all_lat = all_latents.to(**f32).contiguous()
        self.hist = hist_latents.to(**f32).contiguous()
        self.null_lat = null_latent.to(**f32).contiguous()
        null_prompts = null_prompt.to(dev).expand(self.F, -1, -1)
        self.ehs = torch.cat([category_prompts.to(dev) if b[2] else null_prompts for b in br], dim=0).contiguous()
        self.hist_real = _u8([b[0] for b in br], dev)
        self.mutual_real = _u8([b[1] if use_mutual_guidance else 0 for b in br], dev)
        # The last two branches of every category-guidance mode differ only in their PROMPT (difashion.py:388-427: category_prompts vs
        # null_prompts over the same history / mutual flags), so their U-Net inputs are identical: the U-Net computes the part of the walk
        # that sees no text state once for the pair (dfh_unet_set_dup_tail).  DFH_CFG_DEDUP=0 turns it off (A/B).
        flags = [(b[0], b[1] if use_mutual_guidance else 0) for b in br]
        self.dup_tail = self.F if (len(br) >= 2 and flags[-1] == flags[-2] and os.environ.get("DFH_CFG_DEDUP", "1") != "0") else 0
        tab, wt = sampling_tables(olists)
        self.tab, self.wt, self.olen = tab.to(dev), wt.to(dev), olists.shape[1]
        self.sched.set_timesteps(num_inference_steps, device=dev)
        self.ts = list(self.sched._timesteps_host)
        self.is_ddim = isinstance(self.sched, DDIMScheduler)
        self.x_in = torch.empty((self.R * self.F, 2 * init_latents.shape[1], S, S), **f32)
        self.mutual_bf = torch.empty((self.F, self.CL), dtype=torch.bfloat16, device=dev)
        self.eps_comb = torch.empty_like(self.latents) if (keep_eps or not self.is_ddim) else None
        self.mutual = self.null_lat.expand(self.F, -1, -1, -1).contiguous()   # stays when mutual guidance is off
        self.scales = (float(cate_scale), float(hist_scale), float(mutual_scale))
        self.one_minus_eta, self.eta = float(1 - eta), float(eta)
        self.ddim_eta, self.generator, self.use_mutual = ddim_eta, generator, use_mutual_guidance
        self.eps_all = None
        if hasattr(self.unet, "pack"):
            self.unet.pack()
        if hasattr(self.unet, "prepare_run"):       # text K / V^T of every block + the schedule's time-embedding rows: once per run
            self.unet.prepare_run(self.ehs, self.ts)
        return self

    @torch.no_grad()
    def step(self, i: int):
        sp = _lib.stream_ptr
        t = self.ts[i]
        lat = self.latents
        if self.use_mutual:
            _lib.call("dfh_mutual_reduce", _lib.ptr(lat), _lib.ptr(self.all_lat), _lib.ptr(self.tab), _lib.ptr(self.wt),
                      _lib.ptr(self.mutual_bf), None, self.F, self.olen, self.CL, sp())
            self.mutual = self.enc.forward_bf16(self.mutual_bf)
        _lib.call("dfh_assemble_input", _lib.ptr(lat), _lib.ptr(self.mutual), _lib.ptr(self.hist), _lib.ptr(self.null_lat),
                  _lib.ptr(self.mutual_real), _lib.ptr(self.hist_real), _lib.ptr(self.x_in), self.R, self.F, self.CL,
                  self.one_minus_eta, self.eta, 0, sp())
        static = getattr(self.unet, "assume_static_weights", None)
        if static is not None:
            self.unet.assume_static_weights = True      # weights cannot change inside the loop: skip the dirty scan
        try:
            if self.dup_tail and hasattr(self.unet, "_native_forward"):
                self.unet._dup_tail_once = self.dup_tail
            self.eps_all = self.unet(self.x_in, t, self.ehs, return_dict=False)[0]
        finally:
            if hasattr(self.unet, "_dup_tail_once"):
                self.unet._dup_tail_once = 0            # one-shot: a forward that raised before consuming the hint must not leave it behind
            if static is not None:
                self.unet.assume_static_weights = static
        sc, sh, sm = self.scales
        if self.is_ddim:
            k = self.sched.step_coef(t, self.ddim_eta)
            noise = None
            if self.ddim_eta > 0:
                noise = torch.randn(lat.shape, generator=self.generator, device=lat.device, dtype=torch.float32)
            _lib.call("dfh_cfg_step", _lib.ptr(self.eps_all), _lib.ptr(lat), _lib.ptr(self.eps_comb), _lib.ptr(noise),
                      lat.numel(), self.mode, sc, sh, sm, C.byref(k), sp())
        else:
            k = _lib.StepCoef()
            k.kind = STEP_NONE
            _lib.call("dfh_cfg_step", _lib.ptr(self.eps_all), None, _lib.ptr(self.eps_comb), None, lat.numel(), self.mode,
                      sc, sh, sm, C.byref(k), sp())
            self.latents = self.sched.step(self.eps_comb.clone(), t, lat, return_dict=False)[0]
        return self.latents


@torch.no_grad()
def sample_outfits(unet, fashion_encoder: MutualEncoder, scheduler, *, olists: torch.Tensor,
                   all_latents: torch.Tensor, init_latents: torch.Tensor, hist_latents: torch.Tensor,
                   null_latent: torch.Tensor, category_prompts: torch.Tensor, null_prompt: torch.Tensor,
                   num_inference_steps: int = 50, cate_scale: float = 12.0, hist_scale: float = 4.0,
                   mutual_scale: float = 5.0, eta: float = 0.1, ddim_eta: float = 0.0,
                   use_history: bool = True, use_mutual_guidance: bool = True, generator=None,
                   taps: Optional[dict] = None, callback: Optional[Callable] = None, vae=None,
                   output_type: str = "latent") -> torch.Tensor:
    """CFG sampler for the F blank slots of ``olists`` (0 = generate).  Returns final latents (F,4,S,S), or -- with
    ``vae`` and ``output_type="image"`` -- the decoded images in [-1, 1] (F,3,8S,8S):
    ``vae.decode(latents / vae.config.scaling_factor)[0]`` as at difashion.py:580.

    ``all_latents`` (bsz*olen,4,S,S): clean latents of every slot (blank slots unused);
    ``hist_latents`` (F,4,S,S): history rows selected for the blank slots; ``category_prompts``
    (F,77,D); ``null_prompt`` (1,77,D).  ``eta`` is the mutual mix weight (args.eta), ``ddim_eta`` the
    scheduler's stochasticity (fashion_generation's ``eta``)."""
    s = OutfitSampler(unet, fashion_encoder, scheduler).prepare(
        olists=olists, all_latents=all_latents, init_latents=init_latents, hist_latents=hist_latents,
        null_latent=null_latent, category_prompts=category_prompts, null_prompt=null_prompt,
        num_inference_steps=num_inference_steps, cate_scale=cate_scale, hist_scale=hist_scale,
        mutual_scale=mutual_scale, eta=eta, ddim_eta=ddim_eta, use_history=use_history,
        use_mutual_guidance=use_mutual_guidance, generator=generator, keep_eps=taps is not None)
    for i, t in enumerate(s.ts):
        s.step(i)
        if taps is not None:
            taps[f"x_in_{i}"] = s.x_in.clone()
            taps[f"eps_{i}"] = s.eps_comb.clone()
            taps[f"unet_out_{i}"] = s.eps_all.clone()
        if callback is not None:
            callback(i, t, s.latents)
    if output_type == "latent":
        return s.latents
    if vae is None:
        raise ValueError("output_type='image' needs the vae")
    return vae.decode(s.latents / vae.config.scaling_factor, return_dict=False)[0]


class _AssembleInput(torch.autograd.Function):
    """x = cat([(1 - eta) * noisy + eta * masked_mutual, masked_hist]) (difashion.py:186-216).  Only the mutual
    condition carries a gradient (to the MutualEncoder): d mutual = eta * dx[:, :C] on rows whose condition is real."""

    @staticmethod
    def forward(ctx, noisy, mutual, hist, null_lat, m_u8, h_u8, eta):
        n, CL = noisy.shape[0], noisy[0].numel()
        x_in = torch.empty((n, 2 * noisy.shape[1]) + tuple(noisy.shape[2:]), dtype=torch.float32, device=noisy.device)
        _lib.call("dfh_assemble_input", _lib.ptr(noisy), _lib.ptr(mutual), _lib.ptr(hist), _lib.ptr(null_lat),
                  _lib.ptr(m_u8), _lib.ptr(h_u8), _lib.ptr(x_in), 1, n, CL, float(1 - eta), float(eta), 1, _lib.stream_ptr())
        ctx.m_u8, ctx.eta, ctx.shape = m_u8, float(eta), tuple(mutual.shape)
        return x_in

    @staticmethod
    def backward(ctx, dx):
        dx = dx.contiguous().float()
        n = dx.shape[0]
        CL = dx[0].numel() // 2
        dm = torch.empty(ctx.shape, dtype=torch.float32, device=dx.device)
        _lib.call("dfh_assemble_bwd", _lib.ptr(dx), _lib.ptr(ctx.m_u8), _lib.ptr(dm), n, CL, ctx.eta, _lib.stream_ptr())
        return None, dm, None, None, None, None, None


class _WeightedMse(torch.autograd.Function):
    """mean over rows of w[row] * mse(pred[row], target[row]) (difashion.py:255-265; w = min-SNR weights or None).
    The upstream gradient is consumed on the device, so ``loss.backward()`` never synchronises the host."""

    @staticmethod
    def forward(ctx, pred, target, w):
        n, L = pred.shape[0], pred[0].numel()
        rows = torch.empty(n, dtype=torch.float32, device=pred.device)
        _lib.call("dfh_mse_rows", _lib.ptr(pred), _lib.ptr(target), _lib.ptr(rows), n, L, _lib.stream_ptr())
        ctx.kept = (pred, target, w)
        return rows.mean() if w is None else (rows * w).mean()

    @staticmethod
    def backward(ctx, g):
        pred, target, w = ctx.kept
        n, L = pred.shape[0], pred[0].numel()
        g = g.contiguous().float()
        dpred = torch.empty_like(pred)
        _lib.call("dfh_mse_bwd", _lib.ptr(pred), _lib.ptr(target), _lib.ptr(w) if w is not None else None, _lib.ptr(dpred),
                  n, L, 1.0, _lib.ptr(g), _lib.stream_ptr())
        return dpred, None, None


def train_forward(unet, fashion_encoder: MutualEncoder, scheduler, *, latents: torch.Tensor, noise: torch.Tensor,
                  timesteps_outfit: torch.Tensor, null_latent: torch.Tensor, hist_latents: torch.Tensor,
                  ehs: torch.Tensor, null_prompt: torch.Tensor, random_p: Optional[torch.Tensor],
                  random_p_cate: Optional[torch.Tensor], olen: int = 4, eta: float = 0.1,
                  mask_ratio: Optional[float] = 0.2, coupling_mask_ratio: float = 0.3,
                  cate_mask_ratio: Optional[float] = 0.2, snr_gamma: Optional[float] = None,
                  use_history: bool = True, use_mutual_guidance: bool = True,
                  dropout_mask: Optional[torch.Tensor] = None, taps: Optional[dict] = None) -> torch.Tensor:
    """Forward half of one training step: the loss of DiFashion.forward (difashion.py:147-267).

    ``timesteps_outfit`` (bsz,): the draw of :154; ``random_p`` / ``random_p_cate``: the torch.rand draws
    of :188 / :236 (masks are derived from them on the host, as plain index bookkeeping);
    ``hist_latents``: rows chosen at :177-184 before masking.

    With grad enabled the returned loss carries an autograd graph of four native nodes (loss, U-Net, input assembly,
    MutualEncoder): ``loss.backward()`` / ``accelerator.backward(loss)`` (train.py:699) runs the HIP backward and adds
    the gradients into ``.grad`` of the U-Net and encoder parameters."""
    dev = latents.device
    if dev.type != "cuda":
        raise _lib.DfhError("train_forward runs on the HIP path only (device tensors required)")
    f32 = dict(dtype=torch.float32, device=dev)
    n = latents.shape[0]
    CL = latents[0].numel()
    S = latents.shape[-1]
    with torch.no_grad():
        lat = latents.to(**f32).contiguous()
        noi = noise.to(**f32).contiguous()
        t = timesteps_outfit.to(dev).repeat_interleave(olen).long()
        noisy = scheduler.add_noise(lat, noi, t)
        null_lat = null_latent.to(**f32).contiguous()
    sp = _lib.stream_ptr
    if use_mutual_guidance:
        tab, wt = _training_tables_on(n, olen, dev)
        mb = torch.empty((n, CL), dtype=torch.bfloat16, device=dev)
        _lib.call("dfh_mutual_reduce", _lib.ptr(noisy), None, _lib.ptr(tab), _lib.ptr(wt),
                  _lib.ptr(mb), None, n, olen, CL, sp())
        mutual = fashion_encoder.forward_bf16(mb, dropout_mask).contiguous()
    else:
        mutual = null_lat.expand(n, -1, -1, -1).contiguous()
    # condition dropout (difashion.py:186-213): per-row "is real" flags
    hist_real = torch.ones(n, dtype=torch.bool, device=dev)
    mutual_real = torch.ones(n, dtype=torch.bool, device=dev)
    if mask_ratio is not None:
        rp = random_p.to(dev)
        if use_history and use_mutual_guidance:
            hist_real = ~(rp < mask_ratio + coupling_mask_ratio)
            mutual_real = ~((rp >= mask_ratio) & (rp < 2 * mask_ratio + coupling_mask_ratio))
        elif use_history:
            hist_real = ~(rp < mask_ratio)
        elif use_mutual_guidance:
            mutual_real = ~(rp < mask_ratio)
    hist = hist_latents.to(**f32).contiguous()
    m_u8, h_u8 = mutual_real.to(torch.uint8), hist_real.to(torch.uint8)
    x_in = _AssembleInput.apply(noisy, mutual, hist, null_lat, m_u8, h_u8, eta)
    states = ehs.to(dev)
    if cate_mask_ratio is not None:     # torch.where, not boolean-mask assignment: the latter synchronises the host (nonzero)
        drop = (random_p_cate.to(dev) < cate_mask_ratio)[:, None, None]
        states = torch.where(drop, null_prompt.to(dev)[0].to(states.dtype), states)
    states = states.contiguous()
    ptype = scheduler.config.prediction_type
    if ptype == "epsilon":
        target = noi
    elif ptype == "v_prediction":
        target = scheduler.get_velocity(lat, noi, t)
    else:
        raise ValueError(f"Unknown prediction type {ptype}")
    pred = unet(x_in, t, states, return_dict=False)[0]
    pred, target = pred.contiguous(), target.contiguous()
    if taps is not None:
        taps.update(x_in=x_in.detach(), timesteps=t, ehs=states, target=target, pred=pred.detach())
    w = None
    if snr_gamma is not None:
        if getattr(scheduler, "_ac_dev", None) is None or scheduler._ac_dev.device != dev:
            scheduler._ac_dev = scheduler.alphas_cumprod.to(dev)
        ac = scheduler._ac_dev                         # min-SNR weights (difashion.py:258-263), looked up on the device
        snr = ((ac ** 0.5)[t] / ((1.0 - ac) ** 0.5)[t]) ** 2
        w = (torch.minimum(snr, torch.full_like(snr, snr_gamma)) / snr).to(**f32).contiguous()
    return _WeightedMse.apply(pred, target, w)
