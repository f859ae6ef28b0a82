#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference glue.  Runs only in the build container
(needs /root/reference); the committed ``*.npz`` files are what travels.

How: ``/root/reference/DiFashion/models/difashion.py`` dies at import (``diffusers`` is absent),
so a ``sys.modules`` stub supplies the *names* it imports -- no arithmetic lives in the stub:
ModelMixin -> nn.Module (+ ``device``), ConfigMixin -> object, register_to_config -> identity,
randn_tensor -> torch.randn.  A ``DiFashion.__new__`` instance then gets
  * the reference's own ``MutualEncoder`` (real class, real forward),
  * ``unet`` / ``noise_scheduler`` = this repo's oracle restatements (oracle/unet_ref.py,
    oracle/sched_ref.py) wrapped to record every call,
  * stand-in ``vae`` (identity: "images" ARE latents, scaling_factor 1), ``text_encoder``
    (embedding-table lookup) and ``tokenizer`` (constant null ids).
Everything the reference *glue* computes around those handles (mutual reduce, masks, input
assembly, CFG stacking/combination, loop order, loss, SNR weights) is therefore the reference's
own code, captured bit-for-bit.

Usage:  python tests/golden/make_golden.py   (writes tests/golden/*.npz)
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/DiFashion"

from oracle import unet_ref, sched_ref  # noqa: E402

GLUE_CFG = unet_ref.UNetConfig(sample_size=16, block_out_channels=(32, 64, 128, 128),
                               cross_attention_dim=64, num_heads=(1, 2, 2, 2))
HID = 32
CATE_NUM = 11
VOCAB = 23


# ----------------------------------------------------------------------------- stub import
def import_reference():
    class ModelMixin(nn.Module):
        @property
        def device(self):
            return torch.device("cpu")

        @property
        def dtype(self):
            return torch.float32

    def register_to_config(f):
        return f

    def randn_tensor(shape, generator=None, device=None, dtype=None):
        return torch.randn(shape, generator=generator, device=device, dtype=dtype)

    class _Placeholder:
        pass

    class PipeOut:
        def __init__(self, images=None, nsfw_content_detected=None):
            self.images = images

    d = types.ModuleType("diffusers")
    d.AutoencoderKL = _Placeholder
    d.UNet2DConditionModel = _Placeholder
    d.PNDMScheduler = _Placeholder
    mods = {
        "diffusers": d,
        "diffusers.image_processor": types.ModuleType("diffusers.image_processor"),
        "diffusers.pipelines": types.ModuleType("diffusers.pipelines"),
        "diffusers.pipelines.stable_diffusion": types.ModuleType("diffusers.pipelines.stable_diffusion"),
        "diffusers.utils": types.ModuleType("diffusers.utils"),
        "diffusers.utils.import_utils": types.ModuleType("diffusers.utils.import_utils"),
        "diffusers.utils.torch_utils": types.ModuleType("diffusers.utils.torch_utils"),
        "diffusers.configuration_utils": types.ModuleType("diffusers.configuration_utils"),
        "diffusers.models": types.ModuleType("diffusers.models"),
        "diffusers.models.modeling_utils": types.ModuleType("diffusers.models.modeling_utils"),
    }
    mods["diffusers.image_processor"].VaeImageProcessor = _Placeholder
    mods["diffusers.pipelines.stable_diffusion"].StableDiffusionPipelineOutput = PipeOut
    iu = mods["diffusers.utils.import_utils"]
    iu.is_xformers_available = lambda: False
    iu.is_accelerate_available = lambda: False
    iu.is_accelerate_version = lambda *a: False
    mods["diffusers.utils.torch_utils"].randn_tensor = randn_tensor
    mods["diffusers.configuration_utils"].ConfigMixin = object
    mods["diffusers.configuration_utils"].register_to_config = register_to_config
    mods["diffusers.models.modeling_utils"].ModelMixin = ModelMixin
    # transformers is only touched for two class names at import time
    tr = types.ModuleType("transformers")
    tr.CLIPTextModel = _Placeholder
    tr.CLIPTokenizer = _Placeholder
    mods["transformers"] = tr
    import PIL.Image  # noqa: F401  (difashion.py:286 names PIL.Image.Image in an annotation)
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    sys.path.insert(0, REF)
    try:
        import importlib
        ref = importlib.import_module("models.difashion")
    finally:
        sys.path.remove(REF)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return ref


# ----------------------------------------------------------------------------- stand-ins
class Cfg(dict):
    __getattr__ = dict.__getitem__


class FakeVAE:
    """encode(x): "images" are latents already; mode() == sample() == x; scaling_factor 1."""
    config = Cfg(scaling_factor=1.0, latent_channels=4, block_out_channels=(1, 1, 1, 1))

    class _Dist:
        def __init__(self, x):
            self.x = x

        def mode(self):
            return self.x.clone()

        def sample(self):
            return self.x.clone()

    def encode(self, x):
        return types.SimpleNamespace(latent_dist=self._Dist(x))

    def requires_grad_(self, f):
        return self


class FakeText:
    dtype = torch.float32

    def __init__(self, table):
        self.table = table

    def __call__(self, ids):
        return (self.table[ids[:, 0]],)


class FakeTok:
    model_max_length = 77

    def __call__(self, texts, padding=None, max_length=77, truncation=True, return_tensors="pt"):
        return types.SimpleNamespace(input_ids=torch.zeros(len(texts), max_length, dtype=torch.long))


class FakeImageProc:
    def postprocess(self, image, output_type="latent", do_denormalize=None):
        return image


class RecUNet(nn.Module):
    """oracle U-Net that records every call's inputs and output."""

    def __init__(self, cfg, params):
        super().__init__()
        self.inner = unet_ref.OracleUNet(cfg, params)
        self.config = self.inner.config
        self.calls = []

    def forward(self, sample, timestep, encoder_hidden_states, return_dict=True):
        out = self.inner(sample, timestep, encoder_hidden_states, return_dict=True).sample
        self.calls.append(dict(x=sample.detach().clone(), t=torch.as_tensor(timestep).detach().clone(),
                               ehs=encoder_hidden_states.detach().clone(), out=out.detach().clone()))
        return self.inner._Out(out) if return_dict else (out,)


class TensorKeyDict(dict):
    """History container whose membership test works for 0-d tensor keys (the reference looks up
    ``cate in history[uid]`` with a tensor ``cate``: difashion.py:180,382 -- with a plain dict that
    is always False).  Used to exercise the non-null history branch."""

    def __contains__(self, k):
        return dict.__contains__(self, int(k))

    def __getitem__(self, k):
        return dict.__getitem__(self, int(k))


def tiny_weights(seed=7):
    return unet_ref.init_params(GLUE_CFG, seed=seed, w_std=0.05, affine_jitter=0.1)


def checksum(params):
    return np.array([float(sum(v.double().sum() for v in params.values())),
                     float(sum((v.double() ** 2).sum() for v in params.values()))])


def build(ref, sched, args, enc_seed=11):
    m = ref.DiFashion.__new__(ref.DiFashion)
    nn.Module.__init__(m)
    m.args = args
    m.noise_scheduler = sched
    m.tokenizer = FakeTok()
    # text "encoder": ids (n,77) -> table[ids[:,0]]: one (77,D) hidden state per first token id
    g = torch.Generator().manual_seed(5)
    D = GLUE_CFG.cross_attention_dim
    m.text_encoder = FakeText(torch.randn(VOCAB, 1, D, generator=g) + 0.1 * torch.randn(1, 77, D, generator=g))
    m.vae = FakeVAE()
    m.vae_scale_factor = 8
    m.unet = RecUNet(GLUE_CFG, tiny_weights())
    torch.manual_seed(enc_seed)
    m.fashion_encoder = ref.MutualEncoder(cate_num=CATE_NUM, cate_emb_size=8, latent_channels=4,
                                          latent_size=GLUE_CFG.sample_size, hid_dim=HID)
    m.fashion_encoder.apply(ref.xavier_normal_initialization)
    with torch.no_grad():
        for mod in m.fashion_encoder.mlp:
            if isinstance(mod, nn.Linear):
                mod.bias.normal_(0, 0.05)
    m.image_processor = FakeImageProc()
    return m


def enc_state(m):
    return {k: v.detach().clone() for k, v in m.fashion_encoder.state_dict().items() if k.startswith("mlp.")}


def npsave(name, **kw):
    out = {}
    for k, v in kw.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# ----------------------------------------------------------------------------- captures
def cap_mutual_encoder(ref):
    """Full-size MutualEncoder (16384 -> 256 -> 16384), eval mode, weights from a seed."""
    torch.manual_seed(3)
    enc = ref.MutualEncoder(cate_num=50, cate_emb_size=64, latent_channels=4, latent_size=64, hid_dim=256)
    enc.apply(ref.xavier_normal_initialization)
    with torch.no_grad():
        enc.mlp[0].bias.normal_(0, 0.05)
        enc.mlp[3].bias.normal_(0, 0.05)
    enc.eval()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(4, 4, 64, 64, generator=g) * 0.7
    with torch.no_grad():
        y = enc(x)
    sd = {k: v for k, v in enc.state_dict().items() if k.startswith("mlp.")}
    npsave("mutual_encoder_full.npz", x=x, y=y, weight_checksum=checksum(sd),
           w1_head=sd["mlp.0.weight"][:2, :8], w2_head=sd["mlp.3.weight"][:2, :8])


def sample_inputs(bsz, olists, seed):
    g = torch.Generator().manual_seed(seed)
    H = GLUE_CFG.sample_size
    images = torch.randn(bsz, 4, 4, H, H, generator=g) * 0.6       # "outfit images" == clean latents
    null_img = torch.randn(4, H, H, generator=g) * 0.3
    cats = torch.randint(1, CATE_NUM, (bsz, 4), generator=g)
    ids = torch.zeros(bsz, 4, 77, dtype=torch.long)
    ids[:, :, 0] = cats + 5
    uids = torch.arange(bsz) + 100
    oids = torch.arange(bsz) + 900
    init = torch.randn(int((olists == 0).sum()), 4, H, H, generator=g)
    hist = {}
    for u in uids.tolist():
        hist[u] = TensorKeyDict({c: torch.randn(4, H, H, generator=g) * 0.5 for c in range(1, CATE_NUM, 2)})
    return images, null_img, cats, ids, uids, oids, init, hist


def cap_sampling(ref):
    cases = {
        # name: (bsz, olists, scales (cate, hist, mutual), steps, scheduler, use_hist, use_mutual)
        "gor_full_ddim10": (1, [[0, 0, 0, 0]], (12.0, 4.0, 5.0), 10, "ddim", True, True),
        "gor_full_ddim50": (1, [[0, 0, 0, 0]], (12.0, 4.0, 5.0), 50, "ddim", True, True),
        "fitb_full_ddim10": (2, [[3, 0, 5, 6], [7, 8, 9, 0]], (12.0, 4.0, 5.0), 10, "ddim", True, True),
        "mix_full_pndm10": (2, [[0, 0, 5, 6], [7, 0, 0, 0]], (12.0, 4.0, 5.0), 10, "pndm", True, True),
        "fitb_cate_hist": (2, [[3, 0, 5, 6], [0, 8, 9, 2]], (7.5, 3.0, 1.0), 6, "ddim", True, True),
        "fitb_cate_mutual": (2, [[3, 0, 5, 6], [0, 8, 9, 2]], (7.5, 1.0, 3.0), 6, "ddim", True, True),
        "fitb_cate": (2, [[3, 0, 5, 6], [0, 8, 9, 2]], (7.5, 1.0, 1.0), 6, "ddim", True, True),
        "fitb_hist": (2, [[3, 0, 5, 6], [0, 8, 9, 2]], (1.0, 3.0, 1.0), 6, "ddim", True, True),
        "fitb_mutual": (2, [[3, 0, 5, 6], [0, 8, 9, 2]], (1.0, 1.0, 3.0), 6, "ddim", True, True),
        "fitb_hist_mutual": (2, [[3, 0, 5, 6], [0, 8, 9, 2]], (1.0, 2.0, 3.0), 6, "ddim", True, True),
        "fitb_none": (2, [[3, 0, 5, 6], [0, 8, 9, 2]], (1.0, 1.0, 1.0), 6, "ddim", True, True),
        "fitb_nohist_flag": (2, [[3, 0, 5, 6], [0, 8, 9, 2]], (12.0, 4.0, 5.0), 6, "ddim", False, True),
        "fitb_nomutual_flag": (2, [[3, 0, 5, 6], [0, 8, 9, 2]], (12.0, 4.0, 5.0), 6, "ddim", True, False),
    }
    for name, (bsz, ol, (sc, sh, sm), steps, sk, uh, um) in cases.items():
        olists = torch.tensor(ol)
        sched = sched_ref.DDIMRef() if sk == "ddim" else sched_ref.PNDMRef()
        args = types.SimpleNamespace(use_history=uh, use_mutual_guidance=um, eta=0.1)
        m = build(ref, sched, args)
        m.eval()
        images, null_img, cats, ids, uids, oids, init, hist = sample_inputs(bsz, olists, seed=sum(map(ord, name)))
        out = m.fashion_generation(uids=uids, oids=oids, input_ids=ids, olists=olists,
                                   outfit_images=images.reshape(bsz * 4, 4, *images.shape[-2:]),
                                   category=cats, history=hist, num_inference_steps=steps,
                                   category_guidance_scale=sc, hist_guidance_scale=sh, mutual_guidance_scale=sm,
                                   null_img=null_img, eta=0.0, init_latents=init, output_type="latent",
                                   return_dict=True)
        final = out[0].images
        calls = m.unet.calls
        fill = torch.nonzero(olists == 0)
        fill_cate = cats[fill[:, 0], fill[:, 1]]
        # rows the glue selected for history (the lookup at difashion.py:379-386, TensorKeyDict semantics)
        hist_sel = torch.stack([hist[int(uids[o])][int(c)] if (uh and int(c) in hist[int(uids[o])]) else null_img
                                for (o, _), c in zip(fill.tolist(), fill_cate)])
        prompts = m.text_encoder(ids[fill[:, 0], fill[:, 1]])[0]
        null_prompt = m.text_encoder(torch.zeros(1, 77, dtype=torch.long))[0]
        last = len(calls) - 1
        rec = dict(olists=olists, all_latents=images.reshape(bsz * 4, 4, *images.shape[-2:]), null_latent=null_img,
                   init_latents=init, hist_sel=hist_sel, category_prompts=prompts, null_prompt=null_prompt,
                   scales=np.array([sc, sh, sm]), steps=steps, sched=sk, use_history=uh, use_mutual=um,
                   timesteps=torch.stack([c["t"] for c in calls]), n_calls=len(calls), final=final,
                   unet_checksum=checksum(tiny_weights()),
                   **{f"enc.{k}": v for k, v in enc_state(m).items()})
        for tag, i in (("0", 0), ("1", 1), ("last", last)):
            rec[f"x_in_{tag}"] = calls[i]["x"]
            rec[f"ehs_rows_{tag}"] = calls[i]["ehs"][:, 0, :4]
            rec[f"unet_out_{tag}"] = calls[i]["out"]
        npsave(f"sample_{name}.npz", **rec)


def cap_training(ref):
    for name, bsz, gamma, pred, train_mode, uh, um in (
            ("b2_mse", 2, None, "epsilon", False, True, True),
            ("b8_snr5", 8, 5.0, "epsilon", False, True, True),
            ("b2_vpred_snr5", 2, 5.0, "v_prediction", False, True, True),
            ("b2_histonly", 2, None, "epsilon", False, True, False),
            ("b2_mutualonly", 2, 5.0, "epsilon", False, False, True),
            ("b2_trainmode", 2, 5.0, "epsilon", True, True, True)):
        sched = sched_ref.DDIMRef(prediction_type=pred)
        args = types.SimpleNamespace(use_history=uh, use_mutual_guidance=um, eta=0.1, snr_gamma=gamma, noise_offset=0)
        m = build(ref, sched, args)
        m.train(train_mode)
        m.unet.eval()
        H = GLUE_CFG.sample_size
        g = torch.Generator().manual_seed(31 + bsz)
        n_items = 40
        img_dataset = torch.randn(n_items, 4, H, H, generator=g) * 0.6
        null_img = torch.randn(4, H, H, generator=g) * 0.3
        outfits = torch.randint(1, n_items, (bsz, 4), generator=g)
        cats = torch.randint(1, CATE_NUM, (bsz, 4), generator=g)
        ids = torch.zeros(bsz, 4, 77, dtype=torch.long)
        ids[:, :, 0] = cats + 5
        uids = torch.arange(bsz) + 100
        hist = {u: TensorKeyDict({c: torch.randn(4, H, H, generator=g) * 0.5 for c in range(1, CATE_NUM, 2)})
                for u in uids.tolist()}
        batch = dict(uids=uids, outfits=outfits, category=cats, input_ids=ids)
        gen = torch.Generator().manual_seed(77)
        masks = {}
        if train_mode:
            drop = m.fashion_encoder.mlp[2]
            drop.register_forward_hook(lambda mod, i, o: masks.__setitem__("drop", (o != 0).float() / 0.9))
        torch.manual_seed(1234 + bsz)
        loss = m(batch, img_dataset, hist, null_img, 0.2, 0.3, 0.2, torch.float32, gen)
        # replay the global-RNG draws the glue made, in its order (difashion.py:147,154)
        torch.manual_seed(1234 + bsz)
        latents = img_dataset[outfits.reshape(-1)]
        noise = torch.randn_like(latents)
        t_outfit = torch.randint(0, 1000, (bsz,))
        gen2 = torch.Generator().manual_seed(77)
        rp = torch.rand(bsz * 4, generator=gen2)
        rp2 = torch.rand(bsz * 4, generator=gen2)
        call = m.unet.calls[0]
        hist_sel = torch.stack([hist[int(uids[i])][int(c)] if (uh and int(c) in hist[int(uids[i])]) else null_img
                                for i in range(bsz) for c in cats[i]])
        ehs = m.text_encoder(ids.reshape(-1, 77))[0]
        null_prompt = m.text_encoder(torch.zeros(1, 77, dtype=torch.long))[0]
        rec = dict(latents=latents, noise=noise, timesteps_outfit=t_outfit, random_p=rp, random_p_cate=rp2,
                   null_latent=null_img, hist_sel=hist_sel, ehs=ehs, null_prompt=null_prompt,
                   x_in=call["x"], timesteps=call["t"], ehs_rows=call["ehs"][:, 0, :4], unet_out=call["out"],
                   loss=loss.detach(), snr_gamma=np.array(np.nan if gamma is None else gamma), pred_type=pred,
                   use_history=uh, use_mutual=um, unet_checksum=checksum(tiny_weights()),
                   **{f"enc.{k}": v for k, v in enc_state(m).items()})
        if train_mode:
            rec["dropout_mask"] = masks["drop"]
        npsave(f"train_{name}.npz", **rec)


def cap_snr(ref):
    sched = sched_ref.DDIMRef()
    m = build(ref, sched, types.SimpleNamespace())
    t = torch.tensor([0, 1, 500, 999])
    npsave("snr_table.npz", timesteps=t, snr=m.compute_snr(t), alphas_cumprod=sched.alphas_cumprod[t])


def cap_plain_dict_history(ref):
    """As-executed behaviour with the reference's own container type: a plain dict keyed by
    python ints never matches a 0-d tensor key, so history always falls back to null_latent."""
    sched = sched_ref.DDIMRef()
    m = build(ref, sched, types.SimpleNamespace(use_history=True, use_mutual_guidance=True, eta=0.1))
    m.eval()
    olists = torch.tensor([[3, 0, 5, 6], [0, 8, 9, 2]])
    images, null_img, cats, ids, uids, oids, init, hist = sample_inputs(2, olists, seed=99)
    plain = {u: dict(h) for u, h in hist.items()}
    m.fashion_generation(uids=uids, oids=oids, input_ids=ids, olists=olists,
                         outfit_images=images.reshape(8, 4, 16, 16), category=cats, history=plain,
                         num_inference_steps=2, category_guidance_scale=12.0, hist_guidance_scale=4.0,
                         mutual_guidance_scale=5.0, null_img=null_img, eta=0.0, init_latents=init,
                         output_type="latent", return_dict=True)
    x0 = m.unet.calls[0]["x"]
    npsave("plain_dict_history.npz", hist_channels_branch0=x0[:2, 4:], null_latent=null_img)


if __name__ == "__main__":
    torch.set_num_threads(8)
    ref = import_reference()
    cap_mutual_encoder(ref)
    cap_snr(ref)
    cap_plain_dict_history(ref)
    cap_training(ref)
    cap_sampling(ref)
