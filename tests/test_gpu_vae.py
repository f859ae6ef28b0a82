"""-m gpu: AutoencoderKL on the HIP path against the fp32 CPU oracle restatement (oracle/vae_ref.py) on
identical weights.  Stated tolerance as for the U-Net: bf16 activations / weights with fp32 accumulation against fp32 end to
end -> relative L2 <= 3e-2 on the decoded image and on the latent moments (observed values are printed)."""
import pytest
import torch

import difashion_amd as da
from oracle import vae_ref
from tests.gpu_util import DEV, rel_err

pytestmark = pytest.mark.gpu
TOL = 3e-2


def hip_vae(cfg, params):
    m = da.AutoencoderKL(block_out_channels=cfg.block_out_channels, sample_size=cfg.sample_size, init_seed=None)
    m.load_state_dict(params)
    return m.to(DEV).eval()


MID_VAE = vae_ref.VAEConfig(block_out_channels=(64, 128, 256, 256), sample_size=64)


@pytest.mark.parametrize("name,cfg,S", [("tiny", vae_ref.TINY_VAE, 32), ("tiny_64", vae_ref.TINY_VAE, 64), ("mid", MID_VAE, 64)])
def test_vae_encode_decode_match_oracle(name, cfg, S):
    params = vae_ref.init_params(cfg, seed=2, w_std=0.05, affine_jitter=0.1)
    m = hip_vae(cfg, params)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(3, 3, S, S, generator=g) * 2 - 1
    z = torch.randn(3, 4, S // 8, S // 8, generator=g)
    with torch.no_grad():
        ref_m = vae_ref.encode_moments(params, cfg, x)
        ref_img = vae_ref.decode(params, cfg, z)
    dist = m.encode(x.to(DEV)).latent_dist
    img = m.decode(z.to(DEV), return_dict=False)[0]
    e_enc, e_dec = rel_err(dist.parameters.cpu(), ref_m), rel_err(img.cpu(), ref_img)
    print(name, f"moments {e_enc:.2e} decode {e_dec:.2e}")
    assert img.shape == ref_img.shape and dist.mean.shape == (3, 4, S // 8, S // 8)
    assert e_enc <= TOL and e_dec <= TOL
    # latent_dist semantics (difashion.py:129 mode, :144 sample)
    assert torch.equal(dist.mode(), dist.mean)
    gen = torch.Generator(device=DEV).manual_seed(1)
    s1 = dist.sample(gen)
    noise = torch.randn(dist.mean.shape, generator=torch.Generator(device=DEV).manual_seed(1), device=DEV)
    torch.testing.assert_close(s1, dist.mean + torch.exp(0.5 * torch.clamp(dist.parameters[:, 4:], -30, 20)) * noise)


@pytest.mark.timeout(900)
def test_sd_vae_full_size_matches_oracle():
    """The real SD VAE shape (83.65 M parameters): one 512x512 image through encode and one 64x64 latent through decode."""
    cfg = vae_ref.SD_VAE
    params = vae_ref.init_params(cfg, seed=0)
    m = hip_vae(cfg, params)
    g = torch.Generator().manual_seed(4)
    x = torch.rand(1, 3, 512, 512, generator=g) * 2 - 1
    z = torch.randn(1, 4, 64, 64, generator=g)
    with torch.no_grad():
        ref_m = vae_ref.encode_moments(params, cfg, x)
        ref_img = vae_ref.decode(params, cfg, z)
    got_m = m.encode(x.to(DEV)).latent_dist.parameters.cpu()
    got_img = m.decode(z.to(DEV)).sample.cpu()
    e_enc, e_dec = rel_err(got_m, ref_m), rel_err(got_img, ref_img)
    print(f"sd vae: moments {e_enc:.2e} decode {e_dec:.2e}")
    assert got_img.shape == (1, 3, 512, 512) and got_m.shape == (1, 8, 64, 64)
    assert e_enc <= TOL and e_dec <= TOL


def test_vae_api_and_errors():
    cfg = vae_ref.TINY_VAE
    m = hip_vae(cfg, vae_ref.init_params(cfg, seed=5))
    assert m.config.scaling_factor == 0.18215 and m.config.latent_channels == 4 and len(m.config.block_out_channels) == 4
    m.requires_grad_(False)                                         # difashion.py:106
    x = torch.rand(2, 3, 32, 32, device=DEV)
    a = m.encode(x).latent_dist.mode()
    b = m.encode(x, return_dict=False)[0].mode()
    assert torch.equal(a, b)
    one = m.encode(x[1:2]).latent_dist.mode()
    assert rel_err(a[1:2], one) < 2e-3                               # batch rows are independent
    img = m.decode(a / m.config.scaling_factor * m.config.scaling_factor).sample
    assert img.shape == (2, 3, 32, 32) and torch.isfinite(img).all()
    with pytest.raises(ValueError):
        m.encode(torch.rand(2, 3, 30, 30, device=DEV))
    with pytest.raises(da.DfhError, match="no CPU fallback"):
        da.AutoencoderKL(block_out_channels=cfg.block_out_channels, init_seed=None).decode(torch.zeros(1, 4, 4, 4))


def test_sample_outfits_decodes_through_the_vae():
    """difashion.py:456-580 end to end on the HIP path: guided sampling of a 4-item outfit, then vae.decode(latents / scale)."""
    from oracle import unet_ref
    from tests.test_gpu_unet import hip_unet
    ucfg = unet_ref.UNetConfig(sample_size=8, block_out_channels=(64, 128, 256, 256), cross_attention_dim=64, num_heads=(2, 2, 2, 2))
    unet = hip_unet(ucfg, unet_ref.init_params(ucfg, seed=1, w_std=0.05), max_batch=16)
    vcfg = vae_ref.TINY_VAE
    vp = vae_ref.init_params(vcfg, seed=6, w_std=0.05)
    vae = hip_vae(vcfg, vp)
    enc = da.MutualEncoder(cate_num=4, cate_emb_size=8, latent_channels=4, latent_size=8, hid_dim=32).to(DEV).eval()
    g = torch.Generator().manual_seed(7)
    rn = lambda *s: torch.randn(*s, generator=g).to(DEV)
    kw = dict(olists=torch.zeros(1, 4, dtype=torch.long), all_latents=rn(4, 4, 8, 8), init_latents=rn(4, 4, 8, 8),
              hist_latents=rn(4, 4, 8, 8) * 0.2, null_latent=rn(4, 8, 8) * 0.2, category_prompts=rn(4, 77, 64),
              null_prompt=rn(1, 77, 64), num_inference_steps=3)
    lat = da.sample_outfits(unet, enc, da.DDIMScheduler(), **kw)
    img = da.sample_outfits(unet, enc, da.DDIMScheduler(), vae=vae, output_type="image", **kw)
    assert img.shape == (4, 3, 64, 64) and torch.isfinite(img).all()
    with torch.no_grad():
        ref = vae_ref.decode(vp, vcfg, lat.cpu() / vcfg.scaling_factor)
    assert rel_err(img.cpu(), ref) <= TOL
