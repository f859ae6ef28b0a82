"""Opportunistic pin of the oracle against the real third-party classes (last bullet).

The reference owns no U-Net / scheduler / VAE arithmetic: it imports ``UNet2DConditionModel``, ``PNDMScheduler`` and
``AutoencoderKL`` from diffusers 0.18.2 (DiFashion/models/difashion.py:10-19, constructed :64, :74-79).  diffusers is
not installed in the build container nor (normally) on the GPU box, so ``oracle/unet_ref.py``, ``oracle/sched_ref.py``
and ``oracle/vae_ref.py`` are restatements from the published architecture -- "parity unpinned".  Whenever a box DOES
have diffusers, these tests load the oracle's state dict into the real classes and demand fp32 agreement to 1e-5
(relative L2) -- the only route from "unpinned" to "pinned".  They run on CPU (no ``gpu`` marker): fp32, tiny configs.
``__graft_entry__.smoke()`` prints whether diffusers was importable, i.e. whether these ran or skipped.
"""
import numpy as np
import pytest
import torch

from oracle import sched_ref, unet_ref, vae_ref
from tests.helpers import rel_err

diffusers = pytest.importorskip("diffusers", reason="diffusers is not installed: oracle parity stays unpinned")

TOL = 1e-5


def _real_unet(cfg: unet_ref.UNetConfig):
    return diffusers.UNet2DConditionModel(
        sample_size=cfg.sample_size, in_channels=cfg.in_channels, out_channels=cfg.out_channels,
        down_block_types=tuple("CrossAttnDownBlock2D" if a else "DownBlock2D" for a in cfg.down_attn),
        up_block_types=tuple("CrossAttnUpBlock2D" if a else "UpBlock2D" for a in cfg.up_attn),
        block_out_channels=tuple(cfg.block_out_channels), layers_per_block=cfg.layers_per_block,
        cross_attention_dim=cfg.cross_attention_dim,
        attention_head_dim=cfg.num_heads if len(set(cfg.num_heads)) > 1 else cfg.num_heads[0],
        use_linear_projection=cfg.use_linear_projection, norm_num_groups=cfg.norm_num_groups, norm_eps=cfg.norm_eps,
        flip_sin_to_cos=True, freq_shift=0, act_fn="silu").eval()


@pytest.mark.parametrize("name", ["tiny_sd15", "tiny_sd2"])
def test_unet_oracle_equals_diffusers(name):
    base = dict(sample_size=16, block_out_channels=(32, 64, 128, 128), cross_attention_dim=64)
    cfg = (unet_ref.UNetConfig(num_heads=(2, 2, 2, 2), **base) if name == "tiny_sd15" else
           unet_ref.UNetConfig(num_heads=(1, 2, 4, 4), use_linear_projection=True, **base))
    p = unet_ref.init_params(cfg, seed=11, w_std=0.05, affine_jitter=0.1)
    real = _real_unet(cfg)
    missing, unexpected = real.load_state_dict(p, strict=False)
    assert not unexpected, f"oracle keys diffusers does not know: {unexpected[:5]}"
    assert not missing, f"diffusers keys the oracle lacks: {missing[:5]}"
    g = torch.Generator().manual_seed(12)
    x = torch.randn(3, cfg.in_channels, cfg.sample_size, cfg.sample_size, generator=g)
    ehs = torch.randn(3, 77, cfg.cross_attention_dim, generator=g)
    for t in (torch.tensor(981), torch.tensor([1, 500, 999])):
        with torch.no_grad():
            want = real(x, t, encoder_hidden_states=ehs, return_dict=False)[0]
            got = unet_ref.unet_forward(p, cfg, x, t, ehs)
        assert rel_err(got, want) <= TOL, f"{name} t={t.tolist()}: {rel_err(got, want):.3e}"


def test_ddim_oracle_equals_diffusers():
    real = diffusers.DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                                   set_alpha_to_one=False, steps_offset=1, num_train_timesteps=1000)
    ref = sched_ref.DDIMRef()
    assert torch.allclose(real.alphas_cumprod, ref.alphas_cumprod, rtol=1e-6, atol=0)
    g = torch.Generator().manual_seed(13)
    x0, noise = torch.randn(4, 4, 8, 8, generator=g), torch.randn(4, 4, 8, 8, generator=g)
    ts = torch.tensor([0, 1, 500, 999])
    assert rel_err(ref.add_noise(x0, noise, ts), real.add_noise(x0, noise, ts)) <= TOL
    assert rel_err(ref.get_velocity(x0, noise, ts), real.get_velocity(x0, noise, ts)) <= TOL
    for n in (10, 50):
        real.set_timesteps(n)
        ref.set_timesteps(n)
        assert np.array_equal(real.timesteps.numpy(), ref.timesteps.numpy())
        xa = xb = torch.randn(2, 4, 8, 8, generator=g)
        for t in real.timesteps:
            eps = torch.randn(2, 4, 8, 8, generator=g)
            xa = real.step(eps, t, xa, eta=0.0, return_dict=False)[0]
            xb = ref.step(eps, t, xb, eta=0.0, return_dict=False)[0]
        assert rel_err(xb, xa) <= TOL
    assert float(real.init_noise_sigma) == float(ref.init_noise_sigma) == 1.0 and ref.order == 1


def test_pndm_oracle_equals_diffusers():
    real = diffusers.PNDMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", skip_prk_steps=True,
                                   set_alpha_to_one=False, steps_offset=1, num_train_timesteps=1000)
    ref = sched_ref.PNDMRef()
    g = torch.Generator().manual_seed(14)
    for n in (10, 50):
        real.set_timesteps(n)
        ref.set_timesteps(n)
        assert np.array_equal(np.asarray(real.timesteps), ref.timesteps.numpy())
        xa = xb = torch.randn(2, 4, 8, 8, generator=g)
        for t in real.timesteps:
            eps = torch.randn(2, 4, 8, 8, generator=g)
            xa = real.step(eps, t, xa, return_dict=False)[0]
            xb = ref.step(eps, t, xb, return_dict=False)[0]
        assert rel_err(xb, xa) <= TOL


def test_vae_oracle_equals_diffusers():
    cfg = vae_ref.TINY_VAE
    real = diffusers.AutoencoderKL(
        in_channels=cfg.in_channels, out_channels=cfg.out_channels, latent_channels=cfg.latent_channels,
        down_block_types=("DownEncoderBlock2D",) * len(cfg.block_out_channels),
        up_block_types=("UpDecoderBlock2D",) * len(cfg.block_out_channels),
        block_out_channels=tuple(cfg.block_out_channels), layers_per_block=cfg.layers_per_block,
        norm_num_groups=cfg.norm_num_groups, sample_size=cfg.sample_size, act_fn="silu").eval()
    p = vae_ref.init_params(cfg, seed=15, w_std=0.05, affine_jitter=0.1)
    missing, unexpected = real.load_state_dict(p, strict=False)
    assert not unexpected and not missing, (missing[:5], unexpected[:5])
    g = torch.Generator().manual_seed(16)
    x = torch.randn(2, 3, cfg.sample_size, cfg.sample_size, generator=g)
    with torch.no_grad():
        dist = real.encode(x).latent_dist
        z = dist.mode()
        assert rel_err(vae_ref.encode(p, cfg, x), z) <= TOL
        assert rel_err(vae_ref.decode(p, cfg, z), real.decode(z, return_dict=False)[0]) <= TOL
