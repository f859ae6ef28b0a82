"""DDIMScheduler / PNDMScheduler with the diffusers call surface the reference glue uses
(SURVEY.md 8b): ``config.num_train_timesteps`` / ``config.prediction_type``
(DiFashion/models/difashion.py:154,:241), ``add_noise`` (:158), ``get_velocity`` (:244),
``alphas_cumprod`` (:270,:639), ``set_timesteps`` + ``timesteps`` (:356-357), ``scale_model_input``
(:472), ``step(eps, t, x, **{eta, generator}, return_dict=False)[0]`` (:569, kwargs discovered by
``inspect.signature`` :665-673), ``order`` (:433,:574), ``init_noise_sigma`` (:632).

The elementwise updates run in libdifashion_hip.so (dfh_cfg_step / dfh_noise_mix); the host side only
derives the per-step scalar coefficients from the fp32 alpha-bar table, with the same fp32 operations
the published diffusers code uses.  Device tensors only: no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Optional

import numpy as np
import torch

from . import _lib
from .unet import FrozenDict

CFG_NONE = 0
STEP_NONE, STEP_DDIM, STEP_LINEAR = -1, 0, 1


class SchedulerOutput(dict):
    __getattr__ = dict.__getitem__


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and t.device.type != "cuda":
            raise _lib.DfhError("scheduler arithmetic runs on the HIP path only: tensors must live on 'cuda'")


class _SchedulerBase:
    order = 1
    init_noise_sigma = 1.0
    config_name = "scheduler_config.json"

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", steps_offset: int = 1, set_alpha_to_one: bool = False,
                 prediction_type: str = "epsilon", clip_sample: bool = False, **unused):
        if beta_schedule != "scaled_linear":
            raise ValueError("only the 'scaled_linear' schedule of the Stable Diffusion configs is supported")
        if clip_sample:
            raise ValueError("clip_sample is not used by the Stable Diffusion scheduler configs")
        self.config = FrozenDict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                 beta_schedule=beta_schedule, steps_offset=steps_offset,
                                 set_alpha_to_one=set_alpha_to_one, prediction_type=prediction_type,
                                 clip_sample=clip_sample)
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self._timesteps_host = self.timesteps.tolist()
        self._dev_tables = {}

    # -- diffusers persistence surface (difashion.py:64)
    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, **kwargs):
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, cls.config_name)) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        cfg.update(kwargs)
        return cls(**cfg)

    def save_pretrained(self, save_directory: str):
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(dict(self.config, _class_name=type(self).__name__), f, indent=2)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _tables(self, device):
        key = str(device)
        if key not in self._dev_tables:
            a = self.alphas_cumprod
            self._dev_tables[key] = ((a ** 0.5).to(device).contiguous(), ((1 - a) ** 0.5).to(device).contiguous())
        return self._dev_tables[key]

    def _mix(self, x0, noise, timesteps, want_velocity):
        _require_cuda(x0, noise)
        if x0.dtype != torch.float32 or noise.dtype != torch.float32:
            raise TypeError("add_noise / get_velocity operate on float32 latents")
        rows = x0.shape[0]
        L = x0[0].numel()
        t = timesteps.to(device=x0.device, dtype=torch.int64).reshape(-1).contiguous()
        if t.numel() != rows:
            raise ValueError("one timestep per row expected")
        sa, sb = self._tables(x0.device)
        out = torch.empty_like(x0)
        x0c, nc = x0.contiguous(), noise.contiguous()
        _lib.call("dfh_noise_mix", _lib.ptr(x0c), _lib.ptr(nc), _lib.ptr(t), _lib.ptr(sa), _lib.ptr(sb),
                  None if want_velocity else _lib.ptr(out), _lib.ptr(out) if want_velocity else None,
                  rows, L, _lib.stream_ptr())
        return out

    def add_noise(self, original_samples, noise, timesteps):
        return self._mix(original_samples, noise, timesteps, False)

    def get_velocity(self, sample, noise, timesteps):
        return self._mix(sample, noise, timesteps, True)

    def _apply(self, coef: _lib.StepCoef, model_output, sample, noise=None):
        _require_cuda(model_output, sample, noise)
        if model_output.dtype != torch.float32 or sample.dtype != torch.float32:
            raise TypeError("scheduler.step operates on float32 latents")
        out = sample.clone().contiguous()
        eps = model_output.contiguous()
        _lib.call("dfh_cfg_step", _lib.ptr(eps), _lib.ptr(out), None, _lib.ptr(noise), eps.numel(), CFG_NONE,
                  1.0, 1.0, 1.0, C.byref(coef), _lib.stream_ptr())
        return out


class DDIMScheduler(_SchedulerBase):
    """diffusers DDIMScheduler (timestep_spacing 'leading'), SURVEY.md Appendix B."""

    def set_timesteps(self, num_inference_steps: int, device=None):
        if num_inference_steps > self.config.num_train_timesteps:
            raise ValueError("num_inference_steps exceeds num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self._timesteps_host = ts.tolist()
        self.timesteps = torch.from_numpy(ts).to(device) if device is not None else torch.from_numpy(ts)

    def step_coef(self, timestep: int, eta: float = 0.0) -> _lib.StepCoef:
        """Scalar coefficients of one DDIM update, in the fp32 arithmetic of the published code."""
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        var = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)
        std = eta * var ** 0.5
        k = _lib.StepCoef()
        k.kind = STEP_DDIM
        k.vpred = 1 if self.config.prediction_type == "v_prediction" else 0
        if self.config.prediction_type not in ("epsilon", "v_prediction"):
            raise ValueError(f"Unknown prediction type {self.config.prediction_type}")
        k.sqrt_a_t = float(a_t ** 0.5)
        k.sqrt_b_t = float(b_t ** 0.5)
        k.sqrt_a_prev = float(a_prev ** 0.5)
        k.dir_coef = float((1 - a_prev - std ** 2) ** 0.5)
        k.std_dev = float(std)
        return k

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False,
             generator=None, variance_noise=None, return_dict: bool = True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' first")
        k = self.step_coef(timestep, eta)
        noise = None
        if eta > 0:
            noise = variance_noise
            if noise is None:
                noise = torch.randn(model_output.shape, generator=generator, device=model_output.device,
                                    dtype=model_output.dtype)
        prev = self._apply(k, model_output, sample, noise)
        return SchedulerOutput(prev_sample=prev) if return_dict else (prev,)


class PNDMScheduler(_SchedulerBase):
    """diffusers PNDMScheduler, PLMS branch (skip_prk_steps=True as in the SD config the reference
    loads at difashion.py:64).  The multistep epsilon blend is a host-ordered sequence of device
    axpy's; the transfer x' = c_x*x - c_e*eps runs in dfh_cfg_step (STEP_LINEAR)."""

    def __init__(self, *a, skip_prk_steps: bool = True, **k):
        super().__init__(*a, **k)
        if not skip_prk_steps:
            raise ValueError("only skip_prk_steps=True (PLMS) is supported")
        self.config["skip_prk_steps"] = True
        self.ets = []
        self.counter = 0
        self.cur_sample = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        base = (np.arange(0, num_inference_steps) * ratio).round() + self.config.steps_offset
        plms = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy().astype(np.int64)
        self._timesteps_host = plms.tolist()
        self.timesteps = torch.from_numpy(plms).to(device) if device is not None else torch.from_numpy(plms)
        self.ets, self.counter, self.cur_sample = [], 0, None

    def _blend(self, terms):
        """sum_i c_i * e_i on device through the HIP linear-step kernel: x' = 1*x - (-c)*e."""
        (c0, e0), rest = terms[0], terms[1:]
        acc = None
        k = _lib.StepCoef()
        k.kind = STEP_LINEAR
        for c, e in [(c0, e0)] + list(rest):
            if acc is None:
                acc = torch.zeros_like(e)
            k.sqrt_a_t, k.sqrt_b_t = 1.0, -float(c)
            acc = self._apply(k, e, acc)
        return acc

    def step(self, model_output, timestep, sample, return_dict: bool = True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' first")
        t = int(timestep)
        ratio = self.config.num_train_timesteps // self.num_inference_steps
        prev_t = t - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_t = t
            t = t + ratio
        e = self.ets
        if len(e) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(e) == 1 and self.counter == 1:
            model_output = self._blend([(0.5, model_output), (0.5, e[-1])])
            sample = self.cur_sample
            self.cur_sample = None
        elif len(e) == 2:
            model_output = self._blend([(1.5, e[-1]), (-0.5, e[-2])])
        elif len(e) == 3:
            model_output = self._blend([(23 / 12, e[-1]), (-16 / 12, e[-2]), (5 / 12, e[-3])])
        else:
            model_output = self._blend([(55 / 24, e[-1]), (-59 / 24, e[-2]), (37 / 24, e[-3]), (-9 / 24, e[-4])])
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        coeff = float((a_prev / a_t) ** 0.5)
        denom = a_t * b_prev ** 0.5 + (a_t * b_t * a_prev) ** 0.5
        c_e = float((a_prev - a_t) / denom)
        k = _lib.StepCoef()
        k.kind = STEP_LINEAR
        if self.config.prediction_type == "v_prediction":
            # diffusers _get_prev_sample: eps = sqrt(a_t) v + sqrt(b_t) x first, then the same transfer -- still linear in (x, v):
            # x' = (coeff - c_e sqrt(b_t)) x - c_e sqrt(a_t) v   (difashion.py:241-247 trains the v target under this setting)
            k.sqrt_a_t = coeff - c_e * float(b_t ** 0.5)
            k.sqrt_b_t = c_e * float(a_t ** 0.5)
        elif self.config.prediction_type == "epsilon":
            k.sqrt_a_t = coeff
            k.sqrt_b_t = c_e
        else:
            raise ValueError(f"prediction_type given as {self.config.prediction_type} must be one of `epsilon` or `v_prediction`")
        prev = self._apply(k, model_output, sample)
        self.counter += 1
        return SchedulerOutput(prev_sample=prev) if return_dict else (prev,)
