"""UNet2DConditionModel on the MI355X HIP path, behind the diffusers call signature.

Mirrors what the reference glue requires of ``self.unet``:
  * ``unet(sample, timestep, encoder_hidden_states[, return_dict])`` -> ``.sample`` / tuple
    (DiFashion/models/difashion.py:249-253, :518-523); ``timestep`` int, 0-d tensor or (B,) tensor;
  * ``.config.sample_size`` / ``.config.in_channels`` and ``register_to_config(in_channels=8)`` (:85,:99,:327);
  * a replaceable ``.conv_in`` nn.Conv2d with ``out_channels/kernel_size/stride/padding/weight`` (:84-93);
  * ``parameters()/state_dict()/load_state_dict()`` with diffusers key names (train.py:508,545-547,586);
  * ``save_pretrained(dir)`` / ``from_pretrained(dir, subfolder=)`` (train.py:524,545; difashion.py:77-79);
  * ``enable_gradient_checkpointing()`` / ``enable_xformers_memory_efficient_attention()`` (train.py:560,
    difashion.py:118) -- accepted; attention here is always the fused flash-style HIP kernel.

Training (train.py:691-716): with grad enabled and parameters that require grad, ``forward`` runs the
saving forward of the native library and hooks one autograd node onto the result; ``loss.backward()``
then runs the native backward walk, which ADDS the parameter gradients straight into ``p.grad`` (views of
one flat fp32 buffer; created zero-filled when ``p.grad is None``) and hands autograd only the gradient
of ``sample`` (the path to the MutualEncoder).  encoder_hidden_states gets no gradient (frozen CLIP).

All arithmetic runs in libdifashion_hip.so (include/difashion_hip.h).  The fp32 ``nn.Parameter``s are
the master weights (optimizers / EMA / checkpoints keep working on them); ``pack()`` converts them to
the bf16 kernel layouts whenever they change.  There is no PyTorch/CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

from . import _lib


class UNet2DConditionOutput:
    def __init__(self, sample: torch.Tensor):
        self.sample = sample


class FrozenDict(dict):
    """diffusers-style config: attribute and mapping access."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class _UNetStep(torch.autograd.Function):
    """One autograd node for the whole U-Net: forward = dfh_unet_forward_train, backward = dfh_unet_backward."""

    @staticmethod
    def forward(ctx, model, sample, t, ehs, anchor):
        ctx.model = model
        ctx.need_dsample = bool(sample.requires_grad)
        ctx.in_dtype = sample.dtype
        return model._native_forward(sample, t, ehs, train=True)

    @staticmethod
    def backward(ctx, d_out):
        d_sample = ctx.model._native_backward(d_out, ctx.need_dsample)
        if d_sample is not None and d_sample.dtype != ctx.in_dtype:
            d_sample = d_sample.to(ctx.in_dtype)
        return None, d_sample, None, None, None


class _Node(nn.Module):
    """Bare container used to rebuild the diffusers module tree from dotted parameter names."""


_DEFAULT_DOWN = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D")
_DEFAULT_UP = ("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D")


# diffusers UNet2DConditionModel config switches the native walk does not implement: the value the walk assumes (= the SD-1.x /
# SD-2-base value).  Anything else would run silently wrong, so the constructor refuses it.
_FIXED_CONFIG = {
    "act_fn": ("silu", "swish"), "upcast_attention": (False, None), "only_cross_attention": (False, None),
    "dual_cross_attention": (False, None), "class_embed_type": (None,), "num_class_embeds": (None,),
    "resnet_time_scale_shift": ("default", None), "flip_sin_to_cos": (True,), "freq_shift": (0,),
    "mid_block_type": ("UNetMidBlock2DCrossAttn", None), "num_attention_heads": (None,), "center_input_sample": (False, None),
    "downsample_padding": (1, None), "mid_block_scale_factor": (1, 1.0, None), "time_embedding_type": ("positional", None),
    "addition_embed_type": (None,), "encoder_hid_dim": (None,), "encoder_hid_dim_type": (None,), "timestep_post_act": (None,),
    "time_cond_proj_dim": (None,), "conv_in_kernel": (3, None), "conv_out_kernel": (3, None), "class_embeddings_concat": (False, None),
    "resnet_skip_time_act": (False, None), "resnet_out_scale_factor": (1.0, 1, None), "cross_attention_norm": (None,),
    "time_embedding_dim": (None,), "time_embedding_act_fn": (None,), "projection_class_embeddings_input_dim": (None,),
    "mid_block_only_cross_attention": (None, False), "addition_embed_type_num_heads": (64, None),
}


def _reject_unsupported_config(extra: dict):
    bad = []
    for k, v in extra.items():
        if k in _FIXED_CONFIG:
            vv = tuple(v) if isinstance(v, list) else v
            ok = _FIXED_CONFIG[k]
            if isinstance(vv, tuple) and k == "only_cross_attention":
                vv = any(vv)
            if vv not in ok:
                bad.append(f"{k}={v!r} (supported: {ok[0]!r})")
    if bad:
        raise _lib.DfhError("UNet2DConditionModel: config not implemented by the HIP walk: " + "; ".join(bad))


class UNet2DConditionModel(nn.Module):
    config_name = "config.json"
    weights_name = "diffusion_pytorch_model.safetensors"

    def __init__(self, sample_size: int = 64, in_channels: int = 4, out_channels: int = 4,
                 down_block_types: Sequence[str] = _DEFAULT_DOWN, up_block_types: Sequence[str] = _DEFAULT_UP,
                 block_out_channels: Sequence[int] = (320, 640, 1280, 1280), layers_per_block: int = 2,
                 cross_attention_dim: int = 768, attention_head_dim: Union[int, Sequence[int]] = 8,
                 use_linear_projection: bool = False, norm_num_groups: int = 32, norm_eps: float = 1e-5,
                 text_len: int = 77, max_batch: int = 16, init_seed: Optional[int] = 0, init_std: float = 0.02,
                 fp8: bool = False, **unused):
        super().__init__()
        _reject_unsupported_config(unused)
        nb = len(block_out_channels)
        if isinstance(attention_head_dim, int):
            attention_head_dim = (attention_head_dim,) * nb
        if nb > _lib.DFH_MAX_BLOCKS:
            raise ValueError(f"at most {_lib.DFH_MAX_BLOCKS} blocks supported")
        expected_up = tuple("CrossAttnUpBlock2D" if "CrossAttn" in d else "UpBlock2D" for d in reversed(tuple(down_block_types)))
        if tuple(up_block_types) != expected_up:
            raise ValueError("up_block_types must mirror down_block_types")
        self.config = FrozenDict(
            sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
            down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
            block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
            cross_attention_dim=cross_attention_dim, attention_head_dim=tuple(attention_head_dim),
            use_linear_projection=bool(use_linear_projection), norm_num_groups=norm_num_groups,
            norm_eps=norm_eps, text_len=text_len)
        self.max_batch = int(max_batch)
        self.fp8 = bool(fp8)             # e4m3 LayerNorm-fed linears (BASELINE configs[4]); inference walk only, not a config key
        self.assume_static_weights = False   # set by the sampler inside its loop: skip the dirty check
        self._ctx = None
        self._ctx_key = None
        self._names = None
        self._buffers_dev = None
        self._packed_sig = None
        self._train_buffers = None      # (arena16t, grad16, grad32, workspace) once a training step has run
        self._train_batch = 0
        self._grad_flat = None
        self._anchor = None
        # parameter tree straight from the C table (single source of truth for names and shapes)
        ctx = self._make_ctx()
        try:
            table = self._table(ctx)
        finally:
            _lib.raw().dfh_unet_destroy(ctx)
        g = torch.Generator(device="cpu")
        if init_seed is not None:
            g.manual_seed(init_seed)
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], 3, 1, 1)
        for name, shape in table:
            is_norm = ".norm" in name or name.startswith("conv_norm_out")
            if name.endswith(".weight") and not is_norm:
                t = torch.randn(shape, generator=g) * init_std
            elif name.endswith(".weight"):
                t = torch.ones(shape)
            else:
                t = torch.zeros(shape)
            if name.startswith("conv_in."):
                getattr(self.conv_in, name.split(".")[1]).data.copy_(t)
                continue
            parts = name.split(".")
            m = self
            for p in parts[:-1]:
                if p not in m._modules:
                    m.add_module(p, _Node())
                m = m._modules[p]
            m.register_parameter(parts[-1], nn.Parameter(t))

    # ------------------------------------------------------------------ config plumbing
    def register_to_config(self, **kwargs):
        self.config.update(kwargs)

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    @property
    def dtype(self) -> torch.dtype:
        return next(self.parameters()).dtype

    def enable_gradient_checkpointing(self):
        return None

    fp8_attention = False

    def enable_fp8(self, on: bool = True, attention: bool = False):
        """BASELINE configs[4]: run every linear / 1x1 convolution of every transformer block -- proj_in, attn1 q|k and v, both
        to_out, attn2 q, the GEGLU pair ff.net.0 / ff.net.2 and proj_out -- in OCP e4m3 on the block-scaled MFMA
        (csrc/gemm_fp8.hip), fp32 accumulation, per-output-channel weight scales.  Every activation operand is quantised by the
        kernel that produces it: LayerNorm (a scale per token), GroupNorm (the normalised value under a static scale, its affine
        folded into proj_in), the attention epilogues (per image, against the maximum of V), the GEGLU / ff.net.2 epilogues (E8M0
        block scales per 32 channels, consumed by the MFMA's scale operand).  The 3x3 convolutions and the attention products stay
        bf16.  ``attention=True`` additionally runs the self-attention products QK^T / PV on the e4m3 MFMA (csrc/attention_fp8.hip, the
        operands quantised inside the kernel with static factors derived from the projection weights): built and parity-tested, but
        slower and less accurate than the bf16 kernels on this model, hence opt-in.  Sampling path only -- the training forward keeps
        bf16.  Takes effect at the next forward (the context is rebuilt)."""
        self.fp8 = bool(on)
        self.fp8_attention = bool(attention) and self.fp8
        return self

    def enable_xformers_memory_efficient_attention(self, *a, **k):
        return None

    def _c_config(self) -> _lib.UNetConfigC:
        cfg = self.config
        in_ch = self.conv_in.weight.shape[1] if hasattr(self, "conv_in") else cfg["in_channels"]
        c = _lib.UNetConfigC()
        c.sample_size = cfg["sample_size"]
        c.in_channels = int(in_ch)
        c.out_channels = cfg["out_channels"]
        c.num_blocks = len(cfg["block_out_channels"])
        for i, v in enumerate(cfg["block_out_channels"]):
            c.block_out_channels[i] = v
            c.num_heads[i] = cfg["attention_head_dim"][i]
            c.down_attn[i] = 1 if "CrossAttn" in cfg["down_block_types"][i] else 0
        c.layers_per_block = cfg["layers_per_block"]
        c.cross_attention_dim = cfg["cross_attention_dim"]
        c.use_linear_projection = int(cfg["use_linear_projection"])
        c.norm_num_groups = cfg["norm_num_groups"]
        c.norm_eps = cfg["norm_eps"]
        c.text_len = cfg["text_len"]
        return c

    def _make_ctx(self):
        c = self._c_config()
        h = C.c_void_p()
        _lib.call("dfh_unet_create", C.byref(c), C.byref(h))
        return h

    @staticmethod
    def _table(ctx):
        lib = _lib.raw()
        out = []
        for i in range(lib.dfh_unet_num_params(ctx)):
            name = lib.dfh_unet_param_name(ctx, i).decode()
            shape = tuple(lib.dfh_unet_param_dim(ctx, i, d) for d in range(lib.dfh_unet_param_ndim(ctx, i)))
            out.append((name, shape))
        return out

    def param_table(self):
        """[(diffusers key, shape)] as the native library enumerates them."""
        ctx = self._make_ctx()
        try:
            return self._table(ctx)
        finally:
            _lib.raw().dfh_unet_destroy(ctx)

    def __del__(self):
        try:
            if self._ctx is not None:
                _lib.raw().dfh_unet_destroy(self._ctx)
        except Exception:
            pass

    # ------------------------------------------------------------------ device state
    def _ensure_ctx(self, batch: int):
        dev = self.device
        if dev.type != "cuda":
            raise _lib.DfhError("UNet2DConditionModel runs only on the MI355X HIP path: move it to 'cuda' (no CPU fallback)")
        if self.dtype != torch.float32:
            raise _lib.DfhError("master parameters must stay fp32 (the kernels pack their own bf16 copies)")
        in_ch = int(self.conv_in.weight.shape[1])
        if tuple(self.conv_in.kernel_size) != (3, 3) or tuple(self.conv_in.padding) != (1, 1) or tuple(self.conv_in.stride) != (1, 1):
            raise _lib.DfhError("conv_in must be a 3x3 / stride 1 / padding 1 convolution")
        if batch > self.max_batch:
            self.max_batch = batch
        key = (in_ch, self.max_batch, dev.index, self.fp8, self.fp8_attention, tuple(sorted((k, str(v)) for k, v in self.config.items())))
        if self._ctx is not None and key == self._ctx_key:
            return
        if self._ctx is not None:
            _lib.raw().dfh_unet_destroy(self._ctx)
            self._ctx = None
        ctx = self._make_ctx()
        lib = _lib.raw()
        table = self._table(ctx)
        params = dict(self.named_parameters())
        for name, shape in table:
            if name not in params or tuple(params[name].shape) != shape:
                lib.dfh_unet_destroy(ctx)
                raise _lib.DfhError(f"parameter {name}: expected shape {shape}, module has "
                                    f"{tuple(params[name].shape) if name in params else None}")
        if self.fp8:
            if self.fp8_attention:
                _lib.call("dfh_unet_enable_fp8_attention", ctx, 1)
            _lib.call("dfh_unet_enable_fp8", ctx)        # before the workspace is planned
        # zero-filled: padded weight columns (conv_in with 4 input channels -> 8) must read as 0
        a16 = torch.zeros(lib.dfh_unet_arena16_bytes(ctx), dtype=torch.uint8, device=dev)
        a32 = torch.zeros(lib.dfh_unet_arena32_bytes(ctx), dtype=torch.uint8, device=dev)
        wsb = lib.dfh_unet_workspace_bytes(ctx, self.max_batch)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        _lib.call("dfh_unet_bind", ctx, _lib.ptr(a16), _lib.ptr(a32), _lib.ptr(ws), wsb, self.max_batch)
        a8 = None
        if self.fp8:
            a8 = torch.zeros(max(256, lib.dfh_unet_arena8_bytes(ctx)), dtype=torch.uint8, device=dev)
            _lib.call("dfh_unet_bind_fp8", ctx, _lib.ptr(a8))
        self._ctx, self._ctx_key = ctx, key
        self._names = [n for n, _ in table]
        self._buffers_dev = (a16, a32, ws) if a8 is None else (a16, a32, ws, a8)
        self._packed_sig = None
        self._train_buffers = None
        self._train_batch = 0

    def _ensure_train(self, batch: int):
        """Bind the training arenas / workspace (sized for ``batch``; grows on demand)."""
        self._ensure_ctx(min(batch, self.max_batch))
        if self._train_buffers is not None and batch <= self._train_batch:
            return
        lib = _lib.raw()
        dev = self.device
        ctx = self._ctx
        a16t = torch.zeros(lib.dfh_unet_arena16t_bytes(ctx), dtype=torch.uint8, device=dev)
        g16 = torch.zeros(lib.dfh_unet_grad16_bytes(ctx), dtype=torch.uint8, device=dev)     # padding between the packed
        g32 = torch.zeros(lib.dfh_unet_grad32_bytes(ctx), dtype=torch.uint8, device=dev)     # matrices travels through the all-reduce
        wsb = lib.dfh_unet_train_workspace_bytes(ctx, batch)
        self._train_buffers = None          # release the previous workspace before allocating the next
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        _lib.call("dfh_unet_bind_train", ctx, _lib.ptr(a16t), _lib.ptr(g16), _lib.ptr(g32), _lib.ptr(ws), wsb, batch)
        self._train_buffers = (a16t, g16, g32, ws)
        self._train_batch = batch
        self._packed_sig = None             # the transposed packs have to be (re)built
        # the gradient un-pack at the end of backward also leaves sum(g^2) of everything it wrote here: the optimizer's clip norm without
        # another pass over the gradients (training.train_step / FusedAdamW.step(presummed=...)); DFH_TRAIN_UNPACK_NORM=0 switches it off (A/B)
        self._grad_sumsq = None
        self.grad_sumsq_valid = False
        if os.environ.get("DFH_TRAIN_UNPACK_NORM", "1") != "0":
            self._grad_sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        # ALWAYS (re-)registered, null included: the native side must never keep a pointer into a tensor this module has dropped
        _lib.call("dfh_unet_grad_sumsq", ctx, _lib.ptr(self._grad_sumsq) if self._grad_sumsq is not None else None)

    def _signature(self, params):
        return tuple((p.data_ptr(), p._version) for p in params) + (_lib.weight_epoch(),)

    def pack(self, force: bool = False):
        """fp32 master parameters -> bf16 kernel layouts (call happens automatically when they change)."""
        self._ensure_ctx(1)
        named = dict(self.named_parameters())
        plist = [named[n] for n in self._names]
        sig = self._signature(plist)
        if not force and sig == self._packed_sig:
            return
        arr = (C.c_void_p * len(plist))(*[p.data_ptr() for p in plist])
        if self._train_buffers is not None:      # plain + transposed packs from ONE read of the masters
            _lib.call("dfh_unet_pack_all", self._ctx, arr, len(plist), _lib.stream_ptr())
        else:
            _lib.call("dfh_unet_pack", self._ctx, arr, len(plist), _lib.stream_ptr())
        self._packed_sig = sig

    # ------------------------------------------------------------------ per-run constants of a sampling loop
    def prepare_run(self, encoder_hidden_states: torch.Tensor, timesteps) -> None:
        """Compute, once, what every step of a sampling run shares (reference difashion.py:340-357 prompt states, :356 / :456
        timestep list): the cross-attention K / V^T of all transformer blocks for ``encoder_hidden_states`` and the time-embedding
        rows for every entry of ``timesteps``.  Afterwards ``forward(sample, t, encoder_hidden_states)`` with the SAME tensor object
        (unmodified) and a scalar ``t`` from ``timesteps`` skips those launches (dfh_unet_forward_cached); any other call takes the
        normal path.  ``end_run()`` (or a weight update / another prepare_run) drops the cache.  Inference only."""
        ehs = encoder_hidden_states.contiguous()
        B = ehs.shape[0]
        self._ensure_ctx(B)                 # grows the context like forward() does: the cache is filled with launches of batch B
        self.pack()
        ts = [float(t) for t in timesteps]
        tdev = torch.tensor(ts, dtype=torch.float32, device=ehs.device)
        nbytes = _lib.raw().dfh_unet_run_cache_bytes(self._ctx, B, len(ts))
        buf = torch.empty(nbytes, dtype=torch.uint8, device=ehs.device)
        dt = {torch.float32: 0, torch.bfloat16: 1}
        _lib.call("dfh_unet_run_cache", self._ctx, _lib.ptr(ehs), dt[ehs.dtype], B, _lib.ptr(tdev), len(ts), _lib.ptr(buf), nbytes,
                  _lib.stream_ptr())
        self._run_cache = dict(buf=buf, ehs=encoder_hidden_states, ehs_ptr=encoder_hidden_states.data_ptr(),
                               ehs_version=encoder_hidden_states._version, batch=B, index={t: i for i, t in reversed(list(enumerate(ts)))},
                               n=len(ts), sig=self._packed_sig, ctx_key=self._ctx_key)

    def end_run(self) -> None:
        self._run_cache = None

    _run_cache = None

    def _cached_index(self, sample, timestep, ehs):
        """Index into the run cache when this call may use it (same text-state tensor, untouched; same batch; scalar timestep of the
        prepared schedule; weights and context unchanged), else None."""
        rc = self._run_cache
        if rc is None:
            return None
        if (ehs is not rc["ehs"] or ehs.data_ptr() != rc["ehs_ptr"] or ehs._version != rc["ehs_version"] or sample.shape[0] != rc["batch"]
                or rc["ctx_key"] != self._ctx_key or rc["sig"] != self._packed_sig):
            return None
        if torch.is_tensor(timestep):
            if timestep.numel() != 1 or timestep.is_cuda:       # a device scalar would cost a sync to look up: normal path
                return None
            timestep = float(timestep)
        return rc["index"].get(float(timestep))

    def workspace_bytes(self) -> int:
        n = 0 if self._buffers_dev is None else sum(b.numel() for b in self._buffers_dev)
        return n + (0 if self._train_buffers is None else sum(b.numel() for b in self._train_buffers))

    # ------------------------------------------------------------------ native calls
    def _native_forward(self, sample, t, ehs, train: bool, cache_index=None):
        B = sample.shape[0]
        cfg = self.config
        if train:
            self._ensure_train(B)
        else:
            self._ensure_ctx(B)
        if not (self.assume_static_weights and self._packed_sig is not None):
            self.pack()
        dt = {torch.float32: 0, torch.bfloat16: 1}
        out = torch.empty((B, cfg["out_channels"], sample.shape[2], sample.shape[3]), dtype=torch.float32, device=sample.device)
        dup = 0 if train else int(getattr(self, "_dup_tail_once", 0) or 0)
        self._dup_tail_once = 0
        if dup and 2 * dup <= B:            # one-shot: the caller vouches that the last `dup` images repeat the inputs of the ones before
            _lib.call("dfh_unet_set_dup_tail", self._ctx, dup)
        rc = self._run_cache
        if (not train and cache_index is not None and rc is not None and rc["sig"] == self._packed_sig and rc["ctx_key"] == self._ctx_key):
            _lib.call("dfh_unet_forward_cached", self._ctx, _lib.ptr(sample), dt[sample.dtype], _lib.ptr(rc["buf"]), B, rc["n"],
                      int(cache_index), _lib.ptr(out), _lib.stream_ptr())
        else:
            _lib.call("dfh_unet_forward_train" if train else "dfh_unet_forward", self._ctx, _lib.ptr(sample), dt[sample.dtype],
                      _lib.ptr(t), _lib.ptr(ehs), dt[ehs.dtype], _lib.ptr(out), B, _lib.stream_ptr())
        if sample.dtype != torch.float32:
            out = out.to(sample.dtype)
        return out

    def grad_views(self):
        """Table-order list of the parameters' gradient tensors, creating missing ones as zero-filled views of one
        flat fp32 buffer (the layout the fused optimizer and the RCCL gradient all-reduce work on)."""
        named = dict(self.named_parameters())
        plist = [named[n] for n in self._names]
        missing = [p for p in plist if p.requires_grad and p.grad is None]
        if missing:
            if self._grad_flat is None or self._grad_flat.device != self.device:
                total = sum((p.numel() + 63) // 64 * 64 for p in plist)
                self._grad_flat = torch.zeros(total, dtype=torch.float32, device=self.device)
                self._grad_slices = {}
                off = 0
                for p in plist:
                    self._grad_slices[id(p)] = (off, p.numel())
                    off += (p.numel() + 63) // 64 * 64
            if len(missing) == sum(1 for p in plist if p.requires_grad):
                self._grad_flat.zero_()
                for p in missing:
                    off, n = self._grad_slices[id(p)]
                    p.grad = self._grad_flat[off:off + n].view(p.shape)
            else:
                for p in missing:
                    off, n = self._grad_slices[id(p)]
                    v = self._grad_flat[off:off + n].view(p.shape)
                    v.zero_()
                    p.grad = v
        return plist

    def _native_backward(self, d_out, need_dsample: bool):
        d_out = d_out.contiguous().float()
        plist = self.grad_views()
        for p in plist:
            if p.grad is not None and (p.grad.dtype != torch.float32 or not p.grad.is_contiguous()):
                raise _lib.DfhError("parameter gradients must be contiguous fp32 tensors")
        arr = (C.c_void_p * len(plist))(*[(p.grad.data_ptr() if (p.requires_grad and p.grad is not None) else None)
                                          for p in plist])
        cfg = self.config
        d_sample = None
        if need_dsample:
            d_sample = torch.empty((d_out.shape[0], int(self.conv_in.weight.shape[1]), cfg["sample_size"], cfg["sample_size"]),
                                   dtype=torch.float32, device=d_out.device)
        # grads_cleared: set by FusedAdamW.zero_grad(lazy_modules=[...]) -- the gradient views hold stale values that this
        # backward may overwrite instead of accumulate into (no 3.4 GB memset, no read-modify-write)
        overwrite = 1 if getattr(self, "grads_cleared", False) else 0
        self.grads_cleared = False
        self.grads_synced = False
        _lib.stamp_grads(plist)           # FusedAdamW.step() updates only parameters whose gradient was written this epoch
        import torch.distributed as tdist
        from .dist import active as _dist_active
        if self.sync_grads_in_backward and _dist_active():
            self._backward_overlapped(d_out, d_sample, arr, len(plist), overwrite, tdist)
            self.grads_synced = True      # training.train_step then all-reduces only what lies outside this module
        else:
            _lib.call("dfh_unet_backward", self._ctx, _lib.ptr(d_out), _lib.ptr(d_sample) if need_dsample else None, arr,
                      len(plist), overwrite, _lib.stream_ptr())
        # sum(g^2) of the gradients as they now stand (valid until something else writes them; consumed by training.train_step)
        ok = self._grad_sumsq is not None and all(p.requires_grad and p.grad is not None for p in plist)
        self.grad_sumsq_valid = ok
        if ok:                            # validity is CHECKABLE: any in-place edit of a gradient (un-scaling, a manual clip) bumps its version
            self._grad_sumsq_stamp = tuple((p.grad.data_ptr(), p.grad._version) for p in plist)
        return d_sample

    # data-parallel: gradients averaged over the ranks INSIDE backward (DDP semantics; set False around the non-final
    # micro-batches of a gradient-accumulation step, like DDP.no_sync(), and reduce once with dist.all_reduce_gradients)
    sync_grads_in_backward = True
    grads_synced = False
    _grad_sumsq = None
    _grad_sumsq_ok = False
    _grad_sumsq_stamp = None

    @property
    def grad_sumsq_valid(self) -> bool:
        """True while ``_grad_sumsq`` (left behind by the backward's gradient un-pack) still describes ``.grad`` of every parameter:
        set by the backward, and withdrawn as soon as any gradient tensor was replaced or edited in place since (its ``_version`` /
        ``data_ptr`` moved: loss-scale un-scaling, ``clip_grad_norm_``, ``grad.mul_``) -- the optimizer then sums the squares itself."""
        if not self._grad_sumsq_ok or self._grad_sumsq is None or self._grad_sumsq_stamp is None:
            return False
        named = dict(self.named_parameters())
        plist = [named[n] for n in self._names]
        if len(plist) != len(self._grad_sumsq_stamp):
            return False
        return all(p.grad is not None and (p.grad.data_ptr(), p.grad._version) == st for p, st in zip(plist, self._grad_sumsq_stamp))

    @grad_sumsq_valid.setter
    def grad_sumsq_valid(self, v: bool):
        self._grad_sumsq_ok = bool(v)
        if not v:
            self._grad_sumsq_stamp = None

    grad_bucket_bytes = 256 << 20         # few, large buckets: xGMI rings are per-link bound, not latency bound
    # wire format of the gradient exchange: "fp32" (default: one RCCL all-reduce per range -- what the reference's DDP does,
    # train.py:611,699) or "bf16" (EXPLICIT opt-in: half the bytes per xGMI link, fp32 accumulation in rank order, dist.exchange_bf16;
    # every gradient is rounded to bf16 on the wire -- tests/rccl_single_rank_worker.py measures the two-step parameter update moving
    # by 3.2e-2 relative under it).  Nothing is exchanged in a single-process run.
    grad_wire_dtype = "fp32"
    measure_comm = False                  # record how long the compute stream waits for the gradient exchange (bench.py --mode train)
    _comm_events = None

    def last_comm_exposed_ms(self):
        """Time the compute stream spent waiting for the side-stream gradient exchange in the last overlapped backward (ms; the
        part of the exchange the backward walk did NOT hide).  None when nothing was measured.  Synchronises."""
        if self._comm_events is None:
            return None
        e0, e1 = self._comm_events
        e1.synchronize()
        return e0.elapsed_time(e1)

    def _backward_overlapped(self, d_out, d_sample, grad_ptrs, count, overwrite, tdist):
        """The backward walk in pieces (dfh_unet_backward_begin / _next / _finish): whenever a range of the packed fp32
        gradient arena is final -- no layer still to run writes into it -- it is all-reduced on a side stream while the
        walk continues on the compute stream; the un-pack into the master ``.grad`` tensors then reads the averages.
        Replaces DDP's per-parameter autograd hooks + 25 MB buckets (train.py:611,699) with ~14 ranges of 256 MB."""
        lib, dev = _lib.raw(), d_out.device
        g16 = self._train_buffers[1].view(torch.float32)
        g32 = self._train_buffers[2].view(torch.float32)
        world = tdist.get_world_size()
        compute = torch.cuda.current_stream(dev)
        if getattr(self, "_comm_stream", None) is None:
            self._comm_stream = torch.cuda.Stream(device=dev)
        comm = self._comm_stream
        sp = _lib.stream_ptr()
        _lib.call_count("dfh_unet_backward_begin", self._ctx, _lib.ptr(d_out), _lib.ptr(d_sample) if d_sample is not None else None,
                        max(1, self.grad_bucket_bytes // 4), sp)
        lo, hi = C.c_size_t(0), C.c_size_t(0)

        def reduce_range(t):
            ev = torch.cuda.Event()
            ev.record(compute)                    # everything the walk has enqueued so far, i.e. the writers of this range
            with torch.cuda.stream(comm):
                comm.wait_event(ev)
                # RCCL: enqueued behind ``comm``, which then waits for the collective -- the host returns at once and goes
                # on enqueuing the next layers on the compute stream.  (gloo, CPU-staged, blocks the host instead.)
                if self.grad_wire_dtype == "bf16":
                    from .dist import exchange_bf16
                    exchange_bf16(t, capacity=max(1, self.grad_bucket_bytes // 4))       # persistent wire buffers: no allocation here
                else:
                    tdist.all_reduce(t, op=tdist.ReduceOp.SUM)
                    t.div_(world)

        while True:
            rc = lib.dfh_unet_backward_next(self._ctx, C.byref(lo), C.byref(hi), sp)
            if rc < 0:
                raise _lib.DfhError(f"dfh_unet_backward_next failed ({rc}): {_lib.last_error()}")
            if rc == 0:
                break
            reduce_range(g16[lo.value:hi.value])
        reduce_range(g32[: g32.numel()])          # bias / affine gradients: complete only now, a few MB
        if self.measure_comm:                     # compute-stream idle time = the part of the exchange the walk did not hide
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(compute)
            compute.wait_stream(comm)
            e1.record(compute)
            self._comm_events = (e0, e1)
        else:
            compute.wait_stream(comm)
        _lib.call("dfh_unet_backward_finish", self._ctx, grad_ptrs, count, overwrite, sp)

    # ------------------------------------------------------------------ forward
    def forward(self, sample: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor,
                return_dict: bool = True, **unused):
        if sample.dim() != 4:
            raise ValueError("sample must be (B, C, H, W)")
        B, Cin, H, W = sample.shape
        cfg = self.config
        if H != cfg["sample_size"] or W != cfg["sample_size"]:
            raise ValueError(f"sample must be {cfg['sample_size']}x{cfg['sample_size']}, got {H}x{W}")
        if Cin != self.conv_in.weight.shape[1]:
            raise ValueError(f"sample has {Cin} channels, conv_in expects {self.conv_in.weight.shape[1]}")
        if tuple(encoder_hidden_states.shape) != (B, cfg["text_len"], cfg["cross_attention_dim"]):
            raise ValueError(f"encoder_hidden_states must be {(B, cfg['text_len'], cfg['cross_attention_dim'])}, "
                             f"got {tuple(encoder_hidden_states.shape)}")
        if torch.is_grad_enabled() and encoder_hidden_states.requires_grad:
            raise NotImplementedError("no gradient is produced for encoder_hidden_states (frozen text states in the reference)")
        self._ensure_ctx(min(B, self.max_batch))
        dev = sample.device
        train = torch.is_grad_enabled() and (sample.requires_grad or any(p.requires_grad for p in self.parameters()))
        idx = None
        if not train and self._run_cache is not None:
            if not (self.assume_static_weights and self._packed_sig is not None):
                self.pack()                        # a weight update since prepare_run changes the signature and drops the cache
            idx = self._cached_index(sample, timestep, encoder_hidden_states)
        # timestep forms of difashion.py:251 ((B,) int64) and :520 (0-d tensor) / python numbers
        if idx is not None:
            t = None                               # the cached time-embedding row of this schedule entry is used instead
        elif not torch.is_tensor(timestep):
            t = torch.full((B,), float(timestep), dtype=torch.float32, device=dev)
        else:
            t = timestep.to(device=dev, dtype=torch.float32).reshape(-1)
            if t.numel() == 1:
                t = t.expand(B)
            t = t.contiguous()
        if t is not None and t.numel() != B:
            raise ValueError("timestep must be a scalar or have one entry per batch row")
        dt = {torch.float32: 0, torch.bfloat16: 1}
        if sample.dtype not in dt or encoder_hidden_states.dtype not in dt:
            raise TypeError("sample / encoder_hidden_states must be float32 or bfloat16")
        sample = sample.contiguous()
        ehs = encoder_hidden_states.contiguous()
        if train:
            if self._anchor is None or self._anchor.device != dev:
                self._anchor = torch.zeros((), device=dev, requires_grad=True)   # makes autograd call the node's backward
            out = _UNetStep.apply(self, sample, t, ehs, self._anchor)
        else:
            out = self._native_forward(sample, t, ehs, train=False, cache_index=idx)
        return UNet2DConditionOutput(out) if return_dict else (out,)

    def debug_tap(self, name: str) -> torch.Tensor:
        """fp32 NCHW copy of a named intermediate of the last forward (parity tests)."""
        cfg = self.config
        boc = cfg["block_out_channels"]
        nb = len(boc)
        S = cfg["sample_size"]
        shapes = {"conv_in": (boc[0], S), "mid": (boc[-1], S >> (nb - 1))}
        for i in range(nb):
            shapes[f"down{i}"] = (boc[i], S >> min(i + 1, nb - 1))
            shapes[f"up{i}"] = (boc[nb - 1 - i], S >> max(nb - 2 - i, 0))
        c, s = shapes[name]
        B = self._last_batch
        out = torch.empty((B, c, s, s), dtype=torch.float32, device=self.device)
        _lib.call("dfh_unet_debug_tap", self._ctx, name.encode(), _lib.ptr(out), out.numel(), _lib.stream_ptr())
        return out

    def __call__(self, *a, **k):
        out = super().__call__(*a, **k)
        self._last_batch = a[0].shape[0] if a else k["sample"].shape[0]
        return out

    # ------------------------------------------------------------------ checkpoints (diffusers directory layout)
    def save_pretrained(self, save_directory: str, **unused):
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        cfg = dict(self.config)
        cfg["in_channels"] = int(self.conv_in.weight.shape[1])
        cfg["_class_name"] = "UNet2DConditionModel"
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(save_directory, self.weights_name))

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, variant: Optional[str] = None, **kwargs):
        from ._ckpt import load_weights
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, cls.config_name)) as f:
            cfg = json.load(f)
        cfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
        cfg.update({k: v for k, v in kwargs.items() if k in ("max_batch",)})
        model = cls(init_seed=None, **cfg)
        # keys the constructor does not know (e.g. the EMA state diffusers' EMAModel.save_pretrained adds) stay in .config
        model.register_to_config(**{k: v for k, v in cfg.items() if k not in model.config and k != "max_batch"})
        model.load_state_dict(load_weights(d, variant))      # .safetensors, else the .bin diffusers 0.18.2 writes by default
        return model
