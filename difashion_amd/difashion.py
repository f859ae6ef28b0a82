"""``DiFashion`` -- the reference's model class (DiFashion/models/difashion.py:48-676) with its two entry points on the MI355X path.

  * ``forward(batch, img_dataset, history, null_img, mask_ratio, coupling_mask_ratio, cate_mask_ratio, weight_dtype, generator)``
    = the training loss (difashion.py:122-267): VAE-encode the outfit images, draw noise / timesteps, select the history
    latents, then ONE call of the fused ``pipeline.train_forward`` (sibling mean, MutualEncoder, condition masks, input
    assembly, U-Net, min-SNR loss -- with the native backward behind ``loss.backward()``);
  * ``fashion_generation(...)`` = the guided sampler (difashion.py:277-616): resolve the blank slots, history rows and
    prompts, then the fused ``pipeline.sample_outfits`` loop and the VAE decode.

The outer bookkeeping (dict lookups, index lists, RNG draws in the reference's order) is plain Python / torch -- plumbing; every
tensor op of the path runs in libdifashion_hip.so.  ``__init__`` takes the components (any ``vae`` / ``text_encoder`` with the
diffusers / transformers call signatures works, e.g. a ``prompts.PromptTable`` in place of the text encoder);
``DiFashion.from_pretrained_pipeline(args, logger, cate_num, device)`` is the reference's constructor (difashion.py:52-120): it
builds them from the sub-folders of a Stable-Diffusion snapshot with this package's classes and widens ``conv_in``.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Callable, Optional

import torch
import torch.nn as nn

from .pipeline import sample_outfits, train_forward


class DiFashion(nn.Module):
    def __init__(self, args, *, vae, unet, fashion_encoder, noise_scheduler, text_encoder=None, tokenizer=None,
                 prompt_table=None, logger=None):
        super().__init__()
        self.args, self.logger = args, logger
        self.vae, self.unet, self.fashion_encoder = vae, unet, fashion_encoder
        self.noise_scheduler, self.text_encoder, self.tokenizer, self.prompt_table = noise_scheduler, text_encoder, tokenizer, prompt_table
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1)
        if hasattr(vae, "requires_grad_"):
            vae.requires_grad_(False)                                    # difashion.py:106
        if text_encoder is None and prompt_table is None:
            raise ValueError("either text_encoder + tokenizer or a prompt_table is needed")

    @classmethod
    def from_pretrained_pipeline(cls, args, logger=None, cate_num: int = 50, device=None, tokenizer=None, prompt_table=None):
        """What the reference's constructor does (difashion.py:52-120, ``DiFashion(args, logger, cate_num, device)``) on the MI355X classes:
        scheduler / text encoder / VAE / U-Net from the sub-folders of ``args.pretrained_model_name_or_path`` (a Stable-Diffusion snapshot
        directory), the U-Net's ``conv_in`` widened to 8 input channels with the pretrained weights in the first four and zeros behind
        (:82-93), a fresh ``MutualEncoder`` with xavier-normal linears (:95-102), the VAE and the text encoder frozen (:106-107).  The
        tokenizer is host-side string processing and stays transformers' ``CLIPTokenizer`` (loaded from ``tokenizer/`` when none is passed)."""
        from .clip import CLIPTextModel
        from .mutual import MutualEncoder
        from .schedulers import PNDMScheduler
        from .unet import UNet2DConditionModel
        from .vae import AutoencoderKL
        root = args.pretrained_model_name_or_path
        info = logger.info if logger is not None else (lambda *a, **k: None)
        info("scheduler <- scheduler/")
        sched = PNDMScheduler.from_pretrained(root, subfolder="scheduler")
        if tokenizer is None and prompt_table is None:
            info("tokenizer <- tokenizer/ (transformers, host side)")
            from transformers import CLIPTokenizer            # host-only; not part of the compute path
            tokenizer = CLIPTokenizer.from_pretrained(root, subfolder="tokenizer", revision=getattr(args, "revision", None))
        info("text encoder <- text_encoder/ (HIP CLIPTextModel)")
        text = CLIPTextModel.from_pretrained(root, subfolder="text_encoder", revision=getattr(args, "revision", None))
        info("VAE <- vae/ (HIP AutoencoderKL)")
        vae = AutoencoderKL.from_pretrained(root, subfolder="vae")
        info("U-Net <- unet/ (HIP UNet2DConditionModel)")
        unet = UNet2DConditionModel.from_pretrained(root, subfolder="unet")
        info("widening conv_in to [latents | history latents]")
        old = unet.conv_in
        unet.register_to_config(in_channels=8)                # [latents, history_latents]
        with torch.no_grad():
            new_conv_in = nn.Conv2d(8, old.out_channels, old.kernel_size, old.stride, old.padding)
            new_conv_in.weight.zero_()
            new_conv_in.weight[:, :old.weight.shape[1]].copy_(old.weight)
            # (the reference copies the weight only: the bias keeps nn.Conv2d's fresh default init, difashion.py:87-92 -- mirrored as is,
            #  same constructor call, so the same torch RNG state gives the same bias)
        unet.conv_in = new_conv_in
        enc = MutualEncoder(cate_num=cate_num, cate_emb_size=args.category_emb_size, latent_channels=vae.config.latent_channels,
                            latent_size=unet.config.sample_size, hid_dim=args.hid_dim)
        for m in enc.modules():                               # xavier_normal_initialization (difashion.py:731-743)
            if isinstance(m, nn.Linear):
                nn.init.xavier_normal_(m.weight.data)
                if m.bias is not None:
                    nn.init.constant_(m.bias.data, 0)
        text.requires_grad_(False)
        if getattr(args, "enable_xformers_memory_efficient_attention", False):
            unet.enable_xformers_memory_efficient_attention()  # accepted no-op: attention is always the fused HIP kernel
        model = cls(args, vae=vae, unet=unet, fashion_encoder=enc, noise_scheduler=sched, text_encoder=text, tokenizer=tokenizer,
                    prompt_table=prompt_table, logger=logger)
        return model.to(device) if device is not None else model

    @property
    def device(self) -> torch.device:
        return next(self.unet.parameters()).device

    # ---- RNG draws, in the reference's order and on the reference's generators (overridable: tests replay recorded draws)
    def _randn_like(self, t):
        return torch.randn_like(t)

    def _randint(self, high, n):
        return torch.randint(0, high, (n,), device=self.device)

    def _rand(self, n, generator):
        return torch.rand(n, device=self.device, generator=generator)

    def _null_prompt(self, length):
        if self.prompt_table is not None:
            return self.prompt_table.null_prompt
        ids = self.tokenizer([""], padding="max_length", max_length=length, truncation=True, return_tensors="pt").input_ids
        return self.text_encoder(ids.to(self.device))[0]

    def _history_rows(self, uids_of_rows, cates, history, null_latent, use_history):
        # the reference's literal membership test (``cate in history[uid]`` with a 0-d tensor ``cate``), so containers that
        # hash tensors by identity fall back to the null latent exactly as they do there
        rows = []
        for uid, cate in zip(uids_of_rows, cates):
            rows.append(history[uid][cate] if (use_history and cate in history[uid]) else null_latent)
        return torch.stack([r.to(self.device) for r in rows])

    # ------------------------------------------------------------------ training loss (difashion.py:122-267)
    def forward(self, batch, img_dataset, history, null_img, mask_ratio, coupling_mask_ratio, cate_mask_ratio,
                weight_dtype=torch.float32, generator=None, taps: Optional[dict] = None):
        uids, outfits, category, input_ids = batch["uids"], batch["outfits"], batch["category"], batch["input_ids"]
        sf = self.vae.config.scaling_factor
        null_latent = self.vae.encode(null_img.unsqueeze(0).to(self.device, weight_dtype)).latent_dist.mode()[0] * sf
        bsz, olen = len(uids), len(outfits[0])
        images = torch.stack([img_dataset[int(iid)] for i in range(bsz) for iid in outfits[i]]).to(self.device)
        latents = self.vae.encode(images.to(weight_dtype)).latent_dist.sample() * sf
        noise = self._randn_like(latents)
        if getattr(self.args, "noise_offset", 0):
            noise = noise + self.args.noise_offset * torch.randn((latents.shape[0], latents.shape[1], 1, 1), device=latents.device)
        timesteps_outfit = self._randint(self.noise_scheduler.config.num_train_timesteps, bsz)
        hist = self._history_rows([uids[i].item() for i in range(bsz) for _ in category[i]],
                                  [c for i in range(bsz) for c in category[i]], history, null_latent, self.args.use_history)
        random_p = self._rand(bsz * olen, generator) if mask_ratio is not None else None
        if self.prompt_table is not None:
            ehs = self.prompt_table.lookup(torch.stack([torch.as_tensor(c) for c in category]))
        else:
            ehs = self.text_encoder(torch.stack([ids for i in range(bsz) for ids in input_ids[i]]).to(self.device))[0]
        null_prompt = self._null_prompt(self.tokenizer.model_max_length if self.tokenizer is not None else ehs.shape[1])
        random_p_cate = self._rand(bsz * olen, generator) if cate_mask_ratio is not None else None
        dropout_mask = None
        if self.fashion_encoder.training and self.args.use_mutual_guidance:    # nn.Dropout(0.1) of the encoder MLP (difashion.py:33)
            hid = self.fashion_encoder.mlp[0].out_features
            dropout_mask = (torch.rand(bsz * olen, hid, device=self.device) >= 0.1).float() / 0.9
        return train_forward(self.unet, self.fashion_encoder, self.noise_scheduler, latents=latents.float(), noise=noise.float(),
                             timesteps_outfit=timesteps_outfit, null_latent=null_latent.float(), hist_latents=hist.float(), ehs=ehs,
                             null_prompt=null_prompt, random_p=random_p, random_p_cate=random_p_cate, olen=olen, eta=self.args.eta,
                             mask_ratio=mask_ratio, coupling_mask_ratio=coupling_mask_ratio, cate_mask_ratio=cate_mask_ratio,
                             snr_gamma=getattr(self.args, "snr_gamma", None), use_history=self.args.use_history,
                             use_mutual_guidance=self.args.use_mutual_guidance, dropout_mask=dropout_mask, taps=taps)

    # ------------------------------------------------------------------ guided sampling (difashion.py:277-616)
    @torch.no_grad()
    def fashion_generation(self, uids=None, oids=None, input_ids=None, olists=None, outfit_images=None, category=None, history=None,
                           height=None, width=None, num_inference_steps: int = 50, category_guidance_scale: float = 7.5,
                           hist_guidance_scale: float = 7.5, mutual_guidance_scale: float = 7.5, null_img=None, eta: float = 0.0,
                           init_latents=None, generator=None, output_type: Optional[str] = "pil", return_dict: bool = True,
                           callback: Optional[Callable] = None, callback_steps: int = 1):
        dev = self.device
        S = self.unet.config.sample_size
        sf = self.vae.config.scaling_factor
        fill_idx = torch.nonzero(olists == 0)
        fill_num = fill_idx.shape[0]
        fill_cate = category[fill_idx[:, 0], fill_idx[:, 1]]
        fill_uids, fill_oids, full_cate = uids[fill_idx[:, 0]], oids[fill_idx[:, 0]], category[fill_idx[:, 0]]
        if self.prompt_table is not None:
            category_prompts = self.prompt_table.lookup(fill_cate)
        else:
            category_prompts = self.text_encoder(input_ids[fill_idx[:, 0], fill_idx[:, 1]].to(dev))[0]
        null_prompt = self._null_prompt(category_prompts.shape[1])
        if init_latents is None:                      # prepare_latents (difashion.py:618-633): pixel height / width // vae_scale_factor
            lh = (height or S * self.vae_scale_factor) // self.vae_scale_factor
            lw = (width or S * self.vae_scale_factor) // self.vae_scale_factor
            init_latents = torch.randn((fill_num, self.vae.config.latent_channels, lh, lw), generator=generator,
                                       device=dev if generator is None or generator.device.type != "cpu" else "cpu").to(dev)
            init_latents = init_latents * self.noise_scheduler.init_noise_sigma
        null_latent = self.vae.encode(null_img.unsqueeze(0).to(dev)).latent_dist.mode()[0] * sf
        hist = self._history_rows([uids[fill_idx[i][0]].item() for i in range(fill_num)], list(fill_cate), history, null_latent,
                                  self.args.use_history)
        imgs = outfit_images.to(dev)
        all_latents = self.vae.encode(imgs.reshape((-1,) + tuple(imgs.shape[-3:]))).latent_dist.mode() * sf
        cb = (lambda i, t, lat: callback(i, t, lat) if i % callback_steps == 0 else None) if callback is not None else None
        latents = sample_outfits(self.unet, self.fashion_encoder, self.noise_scheduler, olists=olists, all_latents=all_latents.float(),
                                 init_latents=init_latents.float(), hist_latents=hist.float(), null_latent=null_latent.float(),
                                 category_prompts=category_prompts, null_prompt=null_prompt, num_inference_steps=num_inference_steps,
                                 cate_scale=category_guidance_scale, hist_scale=hist_guidance_scale, mutual_scale=mutual_guidance_scale,
                                 eta=self.args.eta, ddim_eta=eta, use_history=self.args.use_history,
                                 use_mutual_guidance=self.args.use_mutual_guidance, generator=generator, callback=cb)
        image = latents if output_type == "latent" else self.vae.decode(latents / sf, return_dict=False)[0]
        image = postprocess(image, output_type)
        if not return_dict:
            results = {}
            for i, uid in enumerate(fill_uids):
                uid, oid = uid.item(), fill_oids[i].item()
                slot = results.setdefault(uid, {}).setdefault(oid, {"images": [], "cates": [], "full_cates": full_cate[i]})
                slot["images"].append(image[i])
                slot["cates"].append(fill_cate[i])
                slot["outfits"] = olists[fill_idx[i][0]]
            return results, init_latents
        return SimpleNamespace(images=image, nsfw_content_detected=None), fill_uids, fill_oids, fill_cate, full_cate, init_latents


def postprocess(image: torch.Tensor, output_type: Optional[str]):
    """diffusers VaeImageProcessor.postprocess with do_denormalize: [-1, 1] -> [0, 1]; "latent"/"pt" tensors, "np" NHWC, "pil"."""
    if output_type == "latent":
        return image
    image = (image / 2 + 0.5).clamp(0, 1)
    if output_type == "pt":
        return image
    arr = image.detach().cpu().permute(0, 2, 3, 1).float().numpy()
    if output_type == "np":
        return arr
    from PIL import Image
    return [Image.fromarray((a * 255).round().astype("uint8")) for a in arr]
