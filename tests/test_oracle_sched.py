"""Closed-form known answers for the scheduler restatement (oracle/sched_ref.py).
diffusers itself is absent, so these anchor the published update rules."""
import math

import numpy as np
import torch

from oracle import sched_ref


def test_beta_schedule_endpoints():
    s = sched_ref.DDIMRef()
    assert abs(float(s.betas[0]) - 0.00085) < 1e-9 and abs(float(s.betas[-1]) - 0.012) < 1e-8
    assert abs(float(s.alphas_cumprod[0]) - (1 - 0.00085)) < 1e-7
    # well-known SD value: alphas_cumprod[999] ~ 0.00466
    assert abs(float(s.alphas_cumprod[999]) - 0.004660) < 2e-5
    assert float(s.final_alpha_cumprod) == float(s.alphas_cumprod[0])


def test_ddim_timesteps_50():
    s = sched_ref.DDIMRef()
    s.set_timesteps(50)
    ts = s.timesteps.tolist()
    assert ts[:3] == [981, 961, 941] and ts[-1] == 1 and len(ts) == 50


def test_pndm_timesteps_50():
    s = sched_ref.PNDMRef()
    s.set_timesteps(50)
    ts = s.timesteps.tolist()
    assert len(ts) == 51 and ts[:4] == [981, 961, 961, 941] and ts[-1] == 1


def test_ddim_step_inverts_add_noise():
    """With the true epsilon, eta=0 DDIM lands exactly on sqrt(a')x0 + sqrt(1-a')eps."""
    s = sched_ref.DDIMRef()
    s.set_timesteps(50)
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    eps = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    s.alphas_cumprod = s.alphas_cumprod.double()
    s.final_alpha_cumprod = s.final_alpha_cumprod.double()
    for t in (981, 501, 21, 1):
        xt = s.add_noise(x0, eps, torch.tensor([t, t]))
        prev = s.step(eps, t, xt, eta=0.0, return_dict=False)[0]
        ap = s.alphas_cumprod[t - 20] if t - 20 >= 0 else s.final_alpha_cumprod
        torch.testing.assert_close(prev, ap.sqrt() * x0 + (1 - ap).sqrt() * eps, rtol=1e-10, atol=1e-10)


def test_ddim_eta_variance():
    s = sched_ref.DDIMRef()
    s.set_timesteps(50)
    x = torch.zeros(1, 4, 4, 4)
    e = torch.zeros(1, 4, 4, 4)
    z = torch.ones(1, 4, 4, 4)
    prev = s.step(e, 501, x, eta=1.0, variance_noise=z, return_dict=False)[0]
    a, ap = float(s.alphas_cumprod[501]), float(s.alphas_cumprod[481])
    sigma = math.sqrt((1 - ap) / (1 - a) * (1 - a / ap))
    np.testing.assert_allclose(prev.numpy(), sigma, rtol=1e-5)


def test_velocity_and_noise_identities():
    s = sched_ref.DDIMRef()
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(3, 4, 4, 4, generator=g)
    n = torch.randn(3, 4, 4, 4, generator=g)
    t = torch.tensor([0, 500, 999])
    xt = s.add_noise(x0, n, t)
    v = s.get_velocity(x0, n, t)
    a = s.alphas_cumprod[t].sqrt()[:, None, None, None]
    b = (1 - s.alphas_cumprod[t]).sqrt()[:, None, None, None]
    torch.testing.assert_close(a * xt - b * v, x0, rtol=1e-4, atol=1e-5)   # v-pred identity
    # v_prediction DDIM step recovers the same x0
    sv = sched_ref.DDIMRef(prediction_type="v_prediction")
    sv.set_timesteps(50)
    out = sv.step(v[1:2], 500 + 1, s.add_noise(x0[1:2], n[1:2], torch.tensor([501])) , return_dict=True)
    assert out["pred_original_sample"].shape == (1, 4, 4, 4)


def test_pndm_first_steps_follow_plms_blend():
    """After the warm-up pair, PLMS uses (3e1 - e2)/2; check the counter logic through 4 steps."""
    s = sched_ref.PNDMRef()
    s.set_timesteps(10)
    x = torch.full((1, 1, 2, 2), 0.5)
    seen = []
    for i, t in enumerate(s.timesteps[:4]):
        e = torch.full_like(x, float(i + 1))
        x = s.step(e, t, x, return_dict=False)[0]
        seen.append(float(x.flatten()[0]))
    assert len(s.ets) == 3 and s.counter == 4
    assert all(np.isfinite(seen))
