"""Optimizer side of the training step on the HIP path (reference: DiFashion/train.py:586-593 AdamW,
:700-704 clip_grad_norm_ / optimizer.step / zero_grad, :707-711 EMAModel.step).

MI355X-first: the parameters of all modules handed to ``FusedAdamW`` are re-homed into ONE flat fp32 buffer, their
gradients into another (the native backward adds into those views), so that a step is three HBM-bound launches
(squared-norm, AdamW with the clip coefficient read on the device, EMA) instead of ~700 x 3 small ones, and the
data-parallel gradient exchange is a single RCCL all-reduce of one buffer (difashion_amd/dist.py).
"""
from __future__ import annotations

import math
import os
from typing import Iterable, List, Optional

import torch

from . import _lib


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction) with clip_grad_norm_ folded in.

    ``max_grad_norm``: when set, ``step()`` first reduces the global squared gradient norm on the device and scales the
    gradients by ``min(1, max_norm / (norm + 1e-6))`` inside the update kernel (accelerator.clip_grad_norm_, train.py:701).
    ``param_groups`` keep working for LR schedulers; every group is a contiguous slice of the flat buffers."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 max_grad_norm: Optional[float] = None, ddp_group="default"):
        """``ddp_group``: the process group whose ranks hold replicas of these parameters.  "default" = the default group when one is
        initialised (the data-parallel training of train.py:611); a ``ProcessGroup`` = that group; ``None`` = this optimizer is NOT
        data-parallel (per-rank independent models): ``step()`` then performs no collective.  With a group, ``step()`` IS a
        collective call (one small all-reduce, see ``_sync_freshness``): every rank of the group must call it the same number of times."""
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.max_grad_norm = max_grad_norm
        self.ddp_group = ddp_group
        plist = [p for g in self.param_groups for p in g["params"]]
        if not plist:
            raise ValueError("no parameters")
        dev = plist[0].device
        if dev.type != "cuda":
            raise _lib.DfhError("FusedAdamW runs on the HIP path only: move the modules to 'cuda' first")
        for p in plist:
            if p.dtype != torch.float32 or p.device != dev:
                raise _lib.DfhError("FusedAdamW needs fp32 parameters on one device")
        total = sum(_align(p.numel()) for p in plist)
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._slices = {}
        self._group_ranges = []
        off = 0
        for g in self.param_groups:
            start = off
            for p in g["params"]:
                n = p.numel()
                view = self.flat_param[off:off + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                gv = self.flat_grad[off:off + n].view(p.shape)
                if p.grad is not None:
                    gv.copy_(p.grad)
                p.grad = gv if p.requires_grad else None
                self._slices[id(p)] = (off, n)
                off += _align(n)
                if p.requires_grad:       # autograd-accumulated gradients (MutualEncoder, anything outside the native U-Net)
                    p.register_post_accumulate_grad_hook(lambda q: setattr(q, "_dfh_grad_epoch", _lib.grad_epoch()))
            self._group_ranges.append((start, off))
        self._step = 0
        self._zero_epoch = _lib.grad_epoch()      # gradients stamped at or after this epoch are fresh
        self._pstep = {}                          # id(param) -> number of updates it has received (torch: state[p]["step"])
        self._all_fresh = False                   # zero_grad(set_to_none=False): torch's "zeros, not None" -> everything updates

    def _fresh(self, p) -> bool:
        """torch.optim.AdamW skips parameters whose ``.grad`` is None.  The gradient views here are never None, so freshness
        is tracked instead: a parameter is updated (weight decay and moments included) only if something wrote its gradient
        since the last ``zero_grad`` -- never-used parameters (``MutualEncoder.category_embedding``, reference
        difashion.py:28), frozen ones, a module whose backward did not run this step (its lazily kept stale gradients must not
        be re-applied) all stay untouched, exactly as with the reference's optimizer."""
        if self._collective_fresh is not None:            # data-parallel: the union over the ranks (see _sync_freshness)
            return id(p) in self._collective_fresh
        return p.requires_grad and (self._all_fresh or getattr(p, "_dfh_grad_epoch", -1) >= self._zero_epoch)

    _collective_fresh = None

    def _sync_freshness(self) -> None:
        """Data-parallel runs: ``step()`` runs after the gradients were AVERAGED over the ranks, so every rank holds the same gradient
        for every parameter -- but the freshness stamps are local (autograd hook / native-backward stamp).  A parameter that received a
        gradient on only some ranks (any data-dependent branch) would be updated only there and the replicas would drift apart
        silently; torch DDP updates it everywhere.  So the fresh set is made collective: one all-reduce(MAX) of a per-parameter mask
        (a few hundred bytes) per step.  No-op without an initialised process group."""
        import torch.distributed as tdist
        self._collective_fresh = None
        if self.ddp_group is None or not (tdist.is_available() and tdist.is_initialized()):
            return
        group = None if isinstance(self.ddp_group, str) else self.ddp_group
        from .dist import active as _dist_active
        if not _dist_active(group):
            return
        plist = [p for g in self.param_groups for p in g["params"]]
        local = [1 if self._fresh(p) else 0 for p in plist]
        dev = self.flat_param.device if tdist.get_backend(group) == "nccl" else torch.device("cpu")
        mask = torch.tensor(local, dtype=torch.int32, device=dev)
        tdist.all_reduce(mask, op=tdist.ReduceOp.MAX, group=group)
        union = mask.cpu().tolist()
        self._collective_fresh = {id(p) for p, f in zip(plist, union) if f and p.requires_grad}

    def mark_fresh(self, params=None) -> None:
        """Declare gradients written by hand (``p.grad.copy_(g)``, a kernel writing into ``flat_grad``) as this step's:
        autograd accumulation and the native U-Net backward stamp their parameters themselves."""
        _lib.stamp_grads(params if params is not None else [p for g in self.param_groups for p in g["params"]])

    def _fresh_ranges(self, a: int, b: int) -> List[tuple]:
        """Merged (start, end, step) ranges of fresh parameters inside the flat range [a, b).  ``step`` is the per-parameter
        update count torch.optim.AdamW keeps in ``state[p]["step"]`` (bias correction): a parameter that gets its first
        gradient late starts at 1 while the others are further on, so a run is split where the counts differ."""
        runs = []
        for g in self.param_groups:
            for p in g["params"]:
                off, n = self._slices[id(p)]
                if off < a or off >= b or not self._fresh(p):
                    continue
                end, st = off + _align(n), self._pstep.get(id(p), 0)
                if runs and runs[-1][1] == off and runs[-1][2] == st:
                    runs[-1][1] = end
                else:
                    runs.append([off, end, st])
        return [tuple(r) for r in runs]

    # the gradient views are permanent: zero_grad never drops them (the native backward accumulates in place)
    def zero_grad(self, set_to_none: bool = True, lazy_modules=()):
        """``lazy_modules``: modules whose native backward can OVERWRITE its gradients (UNet2DConditionModel): their slices
        are not zero-filled here, the module is told that its gradient views hold stale values (``grads_cleared``) and the
        next backward stores instead of accumulating -- 3.4 GB less memset and 3.4 GB less read-modify-write per step."""
        lazy = set()
        for mod in lazy_modules:
            if all(p.grad is not None and id(p) in self._slices for p in mod.parameters() if p.requires_grad):
                lazy.update(id(p) for p in mod.parameters())
                mod.grads_cleared = True
        self._zero_epoch = _lib.next_grad_epoch()
        self._all_fresh = not set_to_none         # the views are kept either way; only the update rule differs (see _fresh)
        if not lazy:
            self.flat_grad.zero_()
        else:
            run = None                      # maximal runs of non-lazy parameters
            for g in self.param_groups:
                for p in g["params"]:
                    off, n = self._slices[id(p)]
                    if id(p) in lazy:
                        if run:
                            self.flat_grad[run[0]:run[1]].zero_()
                            run = None
                    else:
                        run = (run[0] if run else off, off + _align(n))
            if run:
                self.flat_grad[run[0]:run[1]].zero_()
        for g in self.param_groups:
            for p in g["params"]:
                if p.requires_grad and (p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * self._slices[id(p)][0]):
                    off, n = self._slices[id(p)]
                    p.grad = self.flat_grad[off:off + n].view(p.shape)

    def ranges_excluding(self, module) -> List[tuple]:
        """Merged [start, end) element ranges of the flat buffers that hold parameters NOT owned by ``module``."""
        own = {id(p) for p in module.parameters()}
        runs = []
        for off, n in sorted(v for k, v in self._slices.items() if k not in own):
            if runs and runs[-1][1] == off:
                runs[-1][1] = off + n
            else:
                runs.append([off, off + n])
        return [tuple(r) for r in runs]

    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm of the gradients at the last step() (device scalar; no host sync)."""
        return self._sumsq.sqrt()[0]

    def _module_range(self, params) -> Optional[tuple]:
        """[lo, hi) of the flat buffers when ``params`` occupy one gap-free run of it, else None."""
        if not params or any(id(p) not in self._slices for p in params):
            return None
        offs = sorted(self._slices[id(p)] for p in params)
        end = offs[0][0]
        for off, n in offs:
            if off != end:
                return None
            end = off + _align(n)
        return offs[0][0], end

    @torch.no_grad()
    def step(self, closure=None, ema: Optional["EMAModel"] = None, presummed=None):
        """``ema``: an EMAModel over a prefix of this optimizer's parameters (same order): its update is folded into the
        AdamW launch for that range (the caller must then NOT call ema.step()).
        ``presummed``: ``(parameters, tensor)`` -- ``tensor[0]`` already holds the sum of the squares of the CURRENT gradients of
        ``parameters`` (UNet2DConditionModel leaves it behind its backward's gradient un-pack): the clip norm then skips that range of
        the flat gradient buffer (3.4 GB less to read per step).  Ignored unless the parameters are one gap-free, fully fresh run."""
        loss = closure() if closure is not None else None
        ema_end, ema_decay = 0, 0.0
        if ema is not None:
            plist = [p for g in self.param_groups for p in g["params"]][:len(ema._sizes)]
            if (ema.flat.device == self.flat_param.device and [p.numel() for p in plist] == ema._sizes
                    and all(p.requires_grad for p in plist)):
                ema.optimization_step += 1
                ema_decay = ema.get_decay(ema.optimization_step)
                ema_end = ema.flat.numel()
            else:
                raise _lib.DfhError("FusedAdamW.step(ema=...): the EMA must cover a prefix of the optimizer's parameters")
        s = _lib.stream_ptr()
        self._step += 1
        self._sync_freshness()
        total = self.flat_grad.numel()
        for g in self.param_groups:            # this update's per-parameter step counts
            for p in g["params"]:
                if self._fresh(p):
                    self._pstep[id(p)] = self._pstep.get(id(p), 0) + 1
        fresh_all = self._fresh_ranges(0, total)
        pre = None
        if presummed is not None and self.max_grad_norm is not None:
            rng = self._module_range(list(presummed[0]))
            if rng is not None and all(p.requires_grad and self._fresh(p) for p in presummed[0]):
                pre = (rng[0], rng[1], presummed[1])
        self._sumsq.zero_()
        for lo, hi, _ in fresh_all:            # the clip norm counts fresh gradients only (clip_grad_norm_ skips grad None);
            pieces = [(lo, hi)] if pre is None else [(a, b) for a, b in ((lo, min(hi, pre[0])), (max(lo, pre[1]), hi)) if b > a]
            for a, b in pieces:
                _lib.call("dfh_sumsq", self.flat_grad.data_ptr() + 4 * a, b - a, _lib.ptr(self._sumsq), s)   # fixed order: deterministic
        if pre is not None:
            self._sumsq.add_(pre[2].to(self._sumsq.device))
        clip = self.max_grad_norm is not None
        for g, (a, b) in zip(self.param_groups, self._group_ranges):
            if b == a:
                continue
            b1, b2 = g["betas"]
            cursor = a
            for lo, hi, pstep in self._fresh_ranges(a, b) + [(b, b, 0)]:
                hyper = (float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]), int(pstep),
                         _lib.ptr(self._sumsq) if clip else None, float(self.max_grad_norm or 0.0))
                # [cursor, lo): not updated -- but an EMA over them still tracks the (unchanged) parameters, as EMAModel.step does
                if lo > cursor and ema_end > cursor:
                    e = min(lo, ema_end)
                    _lib.call("dfh_ema", ema.flat.data_ptr() + 4 * cursor, self.flat_param.data_ptr() + 4 * cursor, e - cursor,
                              float(ema_decay), s)
                cursor = hi
                fe = min(hi, max(lo, ema_end))         # [lo, fe): fused with the EMA update, [fe, hi): plain
                if fe > lo:
                    _lib.call("dfh_adamw_ema", self.flat_param.data_ptr() + 4 * lo, self.flat_grad.data_ptr() + 4 * lo,
                              self.exp_avg.data_ptr() + 4 * lo, self.exp_avg_sq.data_ptr() + 4 * lo, ema.flat.data_ptr() + 4 * lo,
                              fe - lo, *hyper, float(ema_decay), s)
                if hi > fe:
                    _lib.call("dfh_adamw", self.flat_param.data_ptr() + 4 * fe, self.flat_grad.data_ptr() + 4 * fe,
                              self.exp_avg.data_ptr() + 4 * fe, self.exp_avg_sq.data_ptr() + 4 * fe, hi - fe, *hyper, s)
        self._collective_fresh = None
        _lib.bump_weight_epoch()          # packed bf16 copies of these weights are stale now
        return loss

    def state_dict(self):
        state, idx = {}, 0
        groups = []
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                off, n = self._slices[id(p)]
                state[idx] = dict(step=torch.tensor(float(self._pstep.get(id(p), 0))), exp_avg=self.exp_avg[off:off + n].view(p.shape).clone(),
                                  exp_avg_sq=self.exp_avg_sq[off:off + n].view(p.shape).clone())
                ids.append(idx)
                idx += 1
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": ids})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        idx = 0
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            for k, v in sg.items():
                if k != "params":
                    g[k] = v
            for p in g["params"]:
                st = sd["state"].get(idx, sd["state"].get(str(idx)))
                if st is not None:
                    off, n = self._slices[id(p)]
                    self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                    self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                    self._pstep[id(p)] = int(float(st["step"]))
                    self._step = max(self._step, self._pstep[id(p)])
                idx += 1


class EMAModel:
    """diffusers.training_utils.EMAModel as the reference uses it (train.py:506-511, :707-711, :517-520): shadow
    parameters updated by ``shadow -= (1 - decay) * (shadow - param)`` with the diffusers decay schedule.  When the live
    parameters sit in a FusedAdamW flat buffer the whole update is one launch."""

    def __init__(self, parameters: Iterable[torch.nn.Parameter], decay: float = 0.9999, min_decay: float = 0.0,
                 update_after_step: int = 0, use_ema_warmup: bool = False, inv_gamma: float = 1.0, power: float = 2 / 3,
                 model_cls=None, model_config=None):
        parameters = list(parameters)
        self.decay, self.min_decay, self.update_after_step = decay, min_decay, update_after_step
        self.use_ema_warmup, self.inv_gamma, self.power = use_ema_warmup, inv_gamma, power
        self.optimization_step = 0
        self.model_cls, self.model_config = model_cls, model_config
        self._sizes = [p.numel() for p in parameters]
        self._shapes = [tuple(p.shape) for p in parameters]
        total = sum(_align(n) for n in self._sizes)
        dev = parameters[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.shadow_params = self._views()
        for s, p in zip(self.shadow_params, parameters):
            s.copy_(p.detach())
        self.temp_stored_params = None

    def _views(self) -> List[torch.Tensor]:
        out, off = [], 0
        for n, shape in zip(self._sizes, self._shapes):
            out.append(self.flat[off:off + n].view(shape))
            off += _align(n)
        return out

    def get_decay(self, optimization_step: int) -> float:
        step = max(0, optimization_step - self.update_after_step - 1)
        if step <= 0:
            return 0.0
        cur = 1 - (1 + step / self.inv_gamma) ** -self.power if self.use_ema_warmup else (1 + step) / (10 + step)
        return max(min(cur, self.decay), self.min_decay)

    def to(self, device=None, dtype=None):
        if device is not None and torch.device(device) != self.flat.device:
            self.flat = self.flat.to(device)
            self.shadow_params = self._views()
        return self

    @torch.no_grad()
    def step(self, parameters: Iterable[torch.nn.Parameter]):
        parameters = list(parameters)
        self.optimization_step += 1
        decay = self.get_decay(self.optimization_step)
        if self.flat.device.type != "cuda":
            raise _lib.DfhError("EMAModel.step runs on the HIP path only: call .to('cuda') first")
        s = _lib.stream_ptr()
        # one launch when the live parameters are laid out like the shadow (FusedAdamW flat buffer, same order)
        p0 = parameters[0]
        base = p0.data_ptr()
        off, contiguous = 0, all(p.requires_grad for p in parameters)
        if contiguous:
            for p, n in zip(parameters, self._sizes):
                if p.data_ptr() != base + 4 * off:
                    contiguous = False
                    break
                off += _align(n)
        if contiguous:
            _lib.call("dfh_ema", _lib.ptr(self.flat), base, self.flat.numel() - (_align(self._sizes[-1]) - self._sizes[-1]),
                      float(decay), s)
            return
        for sh, p in zip(self.shadow_params, parameters):
            if p.requires_grad:
                _lib.call("dfh_ema", _lib.ptr(sh), _lib.ptr(p.data), p.numel(), float(decay), s)
            else:
                sh.copy_(p.data)

    @torch.no_grad()
    def copy_to(self, parameters: Iterable[torch.nn.Parameter]):
        for s, p in zip(self.shadow_params, list(parameters)):
            p.data.copy_(s)
        _lib.bump_weight_epoch()

    def store(self, parameters):
        self.temp_stored_params = [p.detach().clone() for p in parameters]

    def restore(self, parameters):
        if self.temp_stored_params is None:
            raise RuntimeError("This ExponentialMovingAverage has no `store()`ed weights to `restore()`")
        for c, p in zip(self.temp_stored_params, parameters):
            p.data.copy_(c)
        self.temp_stored_params = None
        _lib.bump_weight_epoch()

    def state_dict(self):
        return dict(decay=self.decay, min_decay=self.min_decay, optimization_step=self.optimization_step,
                    update_after_step=self.update_after_step, use_ema_warmup=self.use_ema_warmup, inv_gamma=self.inv_gamma,
                    power=self.power, shadow_params=[s.clone() for s in self.shadow_params])

    def load_state_dict(self, sd):
        for k in ("decay", "min_decay", "optimization_step", "update_after_step", "use_ema_warmup", "inv_gamma", "power"):
            if k in sd:
                setattr(self, k, sd[k])
        if sd.get("shadow_params") is not None:
            for s, v in zip(self.shadow_params, sd["shadow_params"]):
                s.copy_(v)

    def save_pretrained(self, path: str):
        if self.model_cls is None:
            raise ValueError("`save_pretrained` can only be used if `model_cls` was defined at __init__.")
        # diffusers layout: the EMA weights as an ordinary model directory, the EMA hyper-state as extra config keys
        cfg = {k: v for k, v in dict(self.model_config).items() if k not in self._STATE_KEYS}
        model = self.model_cls.from_config(cfg) if hasattr(self.model_cls, "from_config") else self.model_cls(init_seed=None, **cfg)
        for s, p in zip(self.shadow_params, model.parameters()):
            p.data.copy_(s.cpu())
        model.register_to_config(**{k: v for k, v in self.state_dict().items() if k != "shadow_params"})
        model.save_pretrained(path)

    _STATE_KEYS = ("decay", "min_decay", "optimization_step", "update_after_step", "use_ema_warmup", "inv_gamma", "power")

    @classmethod
    def from_pretrained(cls, path: str, model_cls) -> "EMAModel":
        """EMAModel.from_pretrained(dir, UNet2DConditionModel) as in the reference's load hook (train.py:530-539)."""
        model = model_cls.from_pretrained(path)
        ema = cls(model.parameters(), model_cls=model_cls, model_config=model.config)
        ema.load_state_dict({k: model.config[k] for k in cls._STATE_KEYS if k in model.config})
        return ema


def clip_grad_norm_(parameters, max_norm: float) -> torch.Tensor:
    """accelerator.clip_grad_norm_ (train.py:701) for gradients that live in arbitrary tensors: device-side norm and
    in-place scale, no host sync.  (FusedAdamW(max_grad_norm=...) folds this into the update instead.)"""
    grads = [p.grad for p in parameters if p.grad is not None]
    ss = torch.zeros(1, dtype=torch.float32, device=grads[0].device)
    for g in grads:
        _lib.call("dfh_sumsq", _lib.ptr(g), g.numel(), _lib.ptr(ss), _lib.stream_ptr())
    norm = ss.sqrt()[0]
    coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
    torch._foreach_mul_(grads, coef)
    return norm


def train_step(unet, fashion_encoder, scheduler, optimizer: FusedAdamW, *, lr_scheduler=None, ema_unet: Optional[EMAModel] = None,
               ema_encoder: Optional[EMAModel] = None, **batch) -> torch.Tensor:
    """One optimisation step of train.py:691-711 on the HIP path: loss (DiFashion.forward) -> native backward ->
    data-parallel gradient all-reduce -> clip + AdamW -> EMA.  ``batch`` are the keyword arguments of
    ``pipeline.train_forward``.  Returns the detached loss (device scalar; nothing here synchronises the host)."""
    from . import dist as _dist
    from .pipeline import train_forward
    loss = train_forward(unet, fashion_encoder, scheduler, **batch)
    loss.backward()
    # the U-Net's backward left sum(g^2) of its gradients behind its un-pack; read its validity NOW, before the exchange of the ranges
    # outside the U-Net edits other slices of the shared flat gradient buffer (views share one version counter)
    sumsq_valid = bool(getattr(unet, "grad_sumsq_valid", False))
    wire = getattr(unet, "grad_wire_dtype", "fp32")       # "bf16" (opt-in): half the bytes per xGMI link, fp32 accumulation
    if getattr(unet, "grads_synced", False):
        # the U-Net averaged its gradients inside backward (overlapped with the walk): reduce what lies outside it
        for lo, hi in optimizer.ranges_excluding(unet):
            _dist.all_reduce_gradients(optimizer.flat_grad[lo:hi], wire=wire)
    else:
        _dist.all_reduce_gradients(optimizer.flat_grad, wire=wire)
    # EMA of the U-Net folded into the AdamW launch when it covers the head of the flat parameter buffer
    first = next(iter(unet.parameters()))
    fuse_ema = (ema_unet is not None and ema_unet.flat.device.type == "cuda"
                and first.data_ptr() == optimizer.flat_param.data_ptr() and all(p.requires_grad for p in unet.parameters()))
    # the U-Net's backward left sum(g^2) of its gradients behind its un-pack: valid if nothing wrote them since (they were averaged over the
    # ranks INSIDE the backward, or there is nothing to average)
    pres = None
    if sumsq_valid and (getattr(unet, "grads_synced", False) or not _dist.active()):
        pres = (list(unet.parameters()), unet._grad_sumsq)
    unet.grad_sumsq_valid = False
    optimizer.step(ema=ema_unet if fuse_ema else None, presummed=pres)
    if lr_scheduler is not None:
        lr_scheduler.step()
    optimizer.zero_grad(lazy_modules=(unet,))
    if ema_unet is not None and not fuse_ema:
        ema_unet.step(unet.parameters())
    if ema_encoder is not None:
        ema_encoder.step(fashion_encoder.parameters())
    return loss.detach()
