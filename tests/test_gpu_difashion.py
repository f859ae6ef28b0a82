"""-m gpu: the ``DiFashion`` class mirror (difashion_amd/difashion.py) driven exactly like the reference's class -- raw batch
dicts, image table, history dicts, generators -- against the golden vectors captured from the REAL class
(tests/golden/make_golden.py).  The raw inputs are re-drawn here from the seeds that script used and checked against the
tensors it recorded (skip if this torch build's CPU RNG stream differs); the stand-in VAE / text encoder are the script's."""
import types

import numpy as np
import pytest
import torch

import difashion_amd as da
from difashion_amd.difashion import DiFashion
from tests.gpu_util import DEV, rel_err
from tests.helpers import GLUE_CFG, glue_unet_params, load
from tests.test_gpu_pipeline import encoder
from tests.test_gpu_unet import hip_unet

pytestmark = pytest.mark.gpu
CATE_NUM, VOCAB, H = 11, 23, GLUE_CFG.sample_size


class Cfg(dict):
    __getattr__ = dict.__getitem__


class IdentityVAE:
    """make_golden's stand-in: "images" are latents already, mode() == sample() == x, scaling_factor 1."""
    config = Cfg(scaling_factor=1.0, latent_channels=4, block_out_channels=(1, 1, 1, 1))

    def encode(self, x):
        d = types.SimpleNamespace(mode=lambda: x.clone(), sample=lambda: x.clone())
        return types.SimpleNamespace(latent_dist=d)

    def decode(self, z, return_dict=False):
        return (z,)


class TableText:
    def __init__(self):
        g = torch.Generator().manual_seed(5)
        D = GLUE_CFG.cross_attention_dim
        self.table = (torch.randn(VOCAB, 1, D, generator=g) + 0.1 * torch.randn(1, 77, D, generator=g)).to(DEV)

    def __call__(self, ids):
        return (self.table[ids[:, 0]],)


class ZeroTok:
    model_max_length = 77

    def __call__(self, texts, padding=None, max_length=77, truncation=True, return_tensors="pt"):
        return types.SimpleNamespace(input_ids=torch.zeros(len(texts), max_length, dtype=torch.long))


class TensorKeyDict(dict):
    def __contains__(self, k):
        return dict.__contains__(self, int(k))

    def __getitem__(self, k):
        return dict.__getitem__(self, int(k))


class Replay(DiFashion):
    """draws from the CPU generators the golden run used (the reference draws on its own device)"""

    def _randn_like(self, t):
        return torch.randn(t.shape).to(t.device)

    def _randint(self, high, n):
        return torch.randint(0, high, (n,)).to(self.device)

    def _rand(self, n, generator):
        return torch.rand(n, generator=generator).to(self.device)


@pytest.fixture(scope="module")
def unet():
    return hip_unet(GLUE_CFG, glue_unet_params(), max_batch=32)


@pytest.mark.parametrize("case,bsz,uh,um", [("b2_mse", 2, True, True), ("b8_snr5", 8, True, True), ("b2_histonly", 2, True, False),
                                           ("b2_mutualonly", 2, False, True)])
def test_difashion_forward_from_raw_batch(case, bsz, uh, um, unet):
    rec = load(f"train_{case}.npz")
    g = torch.Generator().manual_seed(31 + bsz)
    img_dataset = torch.randn(40, 4, H, H, generator=g) * 0.6
    null_img = torch.randn(4, H, H, generator=g) * 0.3
    outfits = torch.randint(1, 40, (bsz, 4), generator=g)
    cats = torch.randint(1, CATE_NUM, (bsz, 4), generator=g)
    ids = torch.zeros(bsz, 4, 77, dtype=torch.long)
    ids[:, :, 0] = cats + 5
    uids = torch.arange(bsz) + 100
    hist = {u: TensorKeyDict({c: torch.randn(4, H, H, generator=g) * 0.5 for c in range(1, CATE_NUM, 2)}) for u in uids.tolist()}
    if not torch.equal(img_dataset[outfits.reshape(-1)], rec["latents"]):
        pytest.skip("torch CPU RNG stream differs from the capture container")
    gamma = float(rec["snr_gamma"])
    args = types.SimpleNamespace(use_history=uh, use_mutual_guidance=um, eta=0.1, snr_gamma=None if np.isnan(gamma) else gamma, noise_offset=0)
    m = Replay(args, vae=IdentityVAE(), unet=unet, fashion_encoder=encoder(rec), noise_scheduler=da.DDIMScheduler(prediction_type=str(rec["pred_type"])),
               text_encoder=TableText(), tokenizer=ZeroTok())
    batch = dict(uids=uids, outfits=outfits, category=cats, input_ids=ids)
    taps = {}
    torch.manual_seed(1234 + bsz)
    with torch.no_grad():
        loss = m(batch, img_dataset, hist, null_img, 0.2, 0.3, 0.2, torch.float32, torch.Generator().manual_seed(77), taps=taps)
    assert torch.equal(taps["timesteps"].cpu(), rec["timesteps"])                       # draws replayed in the reference's order
    torch.testing.assert_close(taps["x_in"].cpu(), rec["x_in"], rtol=0, atol=2e-3)      # image gather, history lookup, masks, assembly
    assert torch.equal(taps["x_in"].cpu()[:, 4:], rec["x_in"][:, 4:])
    assert torch.equal(taps["ehs"].cpu()[:, 0, :4], rec["ehs_rows"])
    print(case, float(loss), float(rec["loss"]))
    assert abs(float(loss) - float(rec["loss"])) <= 2e-2 * abs(float(rec["loss"]))


def sample_inputs(bsz, olists, seed):
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(bsz, 4, 4, H, H, generator=g) * 0.6
    null_img = torch.randn(4, H, H, generator=g) * 0.3
    cats = torch.randint(1, CATE_NUM, (bsz, 4), generator=g)
    ids = torch.zeros(bsz, 4, 77, dtype=torch.long)
    ids[:, :, 0] = cats + 5
    uids, oids = torch.arange(bsz) + 100, torch.arange(bsz) + 900
    init = torch.randn(int((olists == 0).sum()), 4, H, H, generator=g)
    hist = {u: TensorKeyDict({c: torch.randn(4, H, H, generator=g) * 0.5 for c in range(1, CATE_NUM, 2)}) for u in uids.tolist()}
    return images, null_img, cats, ids, uids, oids, init, hist


@pytest.mark.parametrize("name,bsz,ol,scales,steps", [("fitb_full_ddim10", 2, [[3, 0, 5, 6], [7, 8, 9, 0]], (12.0, 4.0, 5.0), 10),
                                                      ("gor_full_ddim10", 1, [[0, 0, 0, 0]], (12.0, 4.0, 5.0), 10),
                                                      ("fitb_cate_hist", 2, [[3, 0, 5, 6], [0, 8, 9, 2]], (7.5, 3.0, 1.0), 6),
                                                      ("mix_full_pndm10", 2, [[0, 0, 5, 6], [7, 0, 0, 0]], (12.0, 4.0, 5.0), 10)])
def test_difashion_fashion_generation_from_raw_inputs(name, bsz, ol, scales, steps, unet):
    rec = load(f"sample_{name}.npz")
    olists = torch.tensor(ol)
    images, null_img, cats, ids, uids, oids, init, hist = sample_inputs(bsz, olists, seed=sum(map(ord, name)))
    if not torch.equal(init, rec["init_latents"]):
        pytest.skip("torch CPU RNG stream differs from the capture container")
    args = types.SimpleNamespace(use_history=True, use_mutual_guidance=True, eta=0.1)
    sched = da.PNDMScheduler() if str(rec["sched"]) == "pndm" else da.DDIMScheduler()       # the reference default is PNDM (difashion.py:64)
    m = DiFashion(args, vae=IdentityVAE(), unet=unet, fashion_encoder=encoder(rec), noise_scheduler=sched,
                  text_encoder=TableText(), tokenizer=ZeroTok())
    d = lambda t: t.to(DEV)
    hist_dev = {u: TensorKeyDict({c: d(v) for c, v in h.items()}) for u, h in hist.items()}
    out = m.fashion_generation(uids=uids, oids=oids, input_ids=ids, olists=olists, outfit_images=d(images.reshape(bsz * 4, 4, H, H)),
                               category=cats, history=hist_dev, num_inference_steps=steps, category_guidance_scale=scales[0],
                               hist_guidance_scale=scales[1], mutual_guidance_scale=scales[2], null_img=d(null_img), eta=0.0,
                               init_latents=d(init), output_type="latent", return_dict=True)
    final, fill_uids, fill_oids, fill_cate, full_cate, init_out = out[0].images, *out[1:]
    fill = torch.nonzero(olists == 0)
    assert torch.equal(fill_uids, uids[fill[:, 0]]) and torch.equal(fill_cate, cats[fill[:, 0], fill[:, 1]])
    assert torch.equal(init_out.cpu(), init)
    e = rel_err(final.cpu(), rec["final"])
    print(name, f"final latents rel err {e:.2e}")
    assert e <= 8e-2
    res, _ = m.fashion_generation(uids=uids, oids=oids, input_ids=ids, olists=olists, outfit_images=d(images.reshape(bsz * 4, 4, H, H)),
                                  category=cats, history=hist_dev, num_inference_steps=2, category_guidance_scale=scales[0],
                                  hist_guidance_scale=scales[1], mutual_guidance_scale=scales[2], null_img=d(null_img),
                                  init_latents=d(init), output_type="pt", return_dict=False)
    uid0, oid0 = int(uids[fill[0, 0]]), int(oids[fill[0, 0]])
    slot = res[uid0][oid0]
    assert set(slot) == {"images", "cates", "full_cates", "outfits"} and slot["images"][0].shape == (4, H, H)
    assert float(slot["images"][0].min()) >= 0.0 and float(slot["images"][0].max()) <= 1.0      # postprocess: [-1,1] -> [0,1]


def test_fashion_generation_draws_its_own_initial_latents(unet):
    """SURVEY.md 8 row a11 (difashion.py:361-371 -> prepare_latents :618-633): called WITHOUT ``init_latents`` the sampler
    draws them from the caller's generator -- ``randn(F, C, h, w) * init_noise_sigma`` -- and returns them.  Golden from the
    real class with the same CPU generator seed (tests/golden/make_golden_autoinit.py): the draw must be bit-identical,
    the trajectory it starts within the sampler tolerance."""
    name = "autoinit_gor_ddim6"
    rec = load(f"sample_{name}.npz")
    olists = torch.tensor([[0, 0, 0, 0], [4, 0, 0, 9]])
    images, null_img, cats, ids, uids, oids, _init, hist = sample_inputs(2, olists, seed=sum(map(ord, name)))
    if not torch.equal(images.reshape(8, 4, H, H), rec["all_latents"]):
        pytest.skip("torch CPU RNG stream differs from the capture container")
    args = types.SimpleNamespace(use_history=True, use_mutual_guidance=True, eta=0.1)
    m = DiFashion(args, vae=IdentityVAE(), unet=unet, fashion_encoder=encoder(rec), noise_scheduler=da.DDIMScheduler(),
                  text_encoder=TableText(), tokenizer=ZeroTok())
    d = lambda t: t.to(DEV)
    hist_dev = {u: TensorKeyDict({c: d(v) for c, v in h.items()}) for u, h in hist.items()}
    sc, sh, sm = (float(v) for v in rec["scales"])
    out = m.fashion_generation(uids=uids, oids=oids, input_ids=ids, olists=olists, outfit_images=d(images.reshape(8, 4, H, H)),
                               category=cats, history=hist_dev, num_inference_steps=int(rec["steps"]), category_guidance_scale=sc,
                               hist_guidance_scale=sh, mutual_guidance_scale=sm, null_img=d(null_img), eta=0.0, init_latents=None,
                               generator=torch.Generator().manual_seed(int(rec["generator_seed"])), output_type="latent",
                               return_dict=True)
    final, init_out = out[0].images, out[-1]
    assert init_out.shape == rec["init_latents"].shape and init_out.device.type == "cuda"
    assert torch.equal(init_out.cpu(), rec["init_latents"])          # same generator, same draw order, * init_noise_sigma
    e = rel_err(final.cpu(), rec["final"])
    print(name, f"final latents rel err {e:.2e}")
    assert e <= 8e-2
    # an explicit height / width (pixels) sets the latent size exactly as prepare_latents does: // vae_scale_factor
    out2 = m.fashion_generation(uids=uids, oids=oids, input_ids=ids, olists=olists, outfit_images=d(images.reshape(8, 4, H, H)),
                                category=cats, history=hist_dev, height=H * m.vae_scale_factor, width=H * m.vae_scale_factor,
                                num_inference_steps=1, category_guidance_scale=sc, hist_guidance_scale=sh, mutual_guidance_scale=sm,
                                null_img=d(null_img), init_latents=None, generator=torch.Generator().manual_seed(int(rec["generator_seed"])),
                                output_type="latent", return_dict=True)
    assert torch.equal(out2[-1].cpu(), rec["init_latents"])
