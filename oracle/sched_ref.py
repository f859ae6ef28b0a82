"""CPU restatement of the scheduler arithmetic on the DiFashion path.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  PARITY UNPINNED vs diffusers
(third-party, diffusers==0.18.2 per reference README.md:27, absent here).
Restates the published DDIM (north star) and PNDM/PLMS (what the reference
instantiates, DiFashion/models/difashion.py:64) update rules, SURVEY.md
Appendix B.  Members mirror what the glue touches:
  add_noise (difashion.py:158), get_velocity (:244), alphas_cumprod (:270,:639),
  set_timesteps / timesteps (:356-357), scale_model_input (:472), step (:569),
  order (:433,:574), init_noise_sigma (:632), config.num_train_timesteps (:154),
  config.prediction_type (:241).
Anchors: closed-form known answers in tests/test_oracle_sched.py.
"""
from __future__ import annotations

import numpy as np
import torch


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class _Base:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 steps_offset=1, set_alpha_to_one=False, prediction_type="epsilon"):
        self.config = _Cfg(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                           beta_end=beta_end, steps_offset=steps_offset,
                           set_alpha_to_one=set_alpha_to_one, prediction_type=prediction_type)
        # "scaled_linear" schedule
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                    dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _coef(self, timesteps, like):
        ac = self.alphas_cumprod.to(device=like.device, dtype=like.dtype)
        a = ac[timesteps] ** 0.5
        s = (1 - ac[timesteps]) ** 0.5
        a = a.flatten()
        s = s.flatten()
        while a.dim() < like.dim():
            a = a.unsqueeze(-1)
            s = s.unsqueeze(-1)
        return a, s

    def add_noise(self, original_samples, noise, timesteps):
        a, s = self._coef(timesteps, original_samples)
        return a * original_samples + s * noise

    def get_velocity(self, sample, noise, timesteps):
        a, s = self._coef(timesteps, sample)
        return a * noise - s * sample


class DDIMRef(_Base):
    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output=False,
             generator=None, variance_noise=None, return_dict: bool = True):
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        if self.config.prediction_type == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif self.config.prediction_type == "v_prediction":
            x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
            eps = a_t ** 0.5 * model_output + b_t ** 0.5 * sample
        else:
            raise ValueError(self.config.prediction_type)
        var = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)
        std = eta * var ** 0.5
        direction = (1 - a_prev - std ** 2) ** 0.5 * eps
        prev = a_prev ** 0.5 * x0 + direction
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            prev = prev + std * variance_noise
        return _Cfg(prev_sample=prev, pred_original_sample=x0) if return_dict else (prev,)


class PNDMRef(_Base):
    """PLMS branch only (skip_prk_steps=True, as in the SD scheduler config)."""

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        base = (np.arange(0, num_inference_steps) * ratio).round() + self.config.steps_offset
        plms = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()
        self.timesteps = torch.from_numpy(plms.astype(np.int64))
        self.ets = []
        self.counter = 0
        self.cur_sample = None

    def _prev(self, sample, t, prev_t, eps):
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        if self.config.prediction_type == "v_prediction":
            eps = a_t ** 0.5 * eps + b_t ** 0.5 * sample
        coeff = (a_prev / a_t) ** 0.5
        denom = a_t * b_prev ** 0.5 + (a_t * b_t * a_prev) ** 0.5
        return coeff * sample - (a_prev - a_t) * eps / denom

    def step(self, model_output, timestep, sample, return_dict: bool = True):
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_t = t
            t = t + self.config.num_train_timesteps // self.num_inference_steps
        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            model_output = (model_output + self.ets[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            model_output = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            model_output = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])
        prev = self._prev(sample, t, prev_t, model_output)
        self.counter += 1
        return _Cfg(prev_sample=prev) if return_dict else (prev,)
