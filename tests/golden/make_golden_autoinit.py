#!/usr/bin/env python3
"""Golden vector for SURVEY.md 8 row a11: the sampler called WITHOUT ``init_latents`` -- the reference then draws them in
``prepare_latents`` (DiFashion/models/difashion.py:361-371, 618-633: ``randn_tensor(shape, generator, device, dtype) *
noise_scheduler.init_noise_sigma``) and returns them as the last element of its result tuple.  Runs only in the build
container (imports the real reference class through tests/golden/make_golden.py's name-only stub); the ``.npz`` travels.

Usage:  python tests/golden/make_golden_autoinit.py   (writes tests/golden/sample_autoinit_gor_ddim6.npz)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from oracle import sched_ref  # noqa: E402

SEED = 4242


def main():
    torch.set_num_threads(8)
    ref = mg.import_reference()
    name, bsz, ol, (sc, sh, sm), steps = "autoinit_gor_ddim6", 2, [[0, 0, 0, 0], [4, 0, 0, 9]], (12.0, 4.0, 5.0), 6
    olists = torch.tensor(ol)
    m = mg.build(ref, sched_ref.DDIMRef(), types.SimpleNamespace(use_history=True, use_mutual_guidance=True, eta=0.1))
    m.eval()
    images, null_img, cats, ids, uids, oids, _unused_init, hist = mg.sample_inputs(bsz, olists, seed=sum(map(ord, name)))
    H = images.shape[-1]
    out = m.fashion_generation(uids=uids, oids=oids, input_ids=ids, olists=olists, outfit_images=images.reshape(bsz * 4, 4, H, H),
                               category=cats, history=hist, num_inference_steps=steps,
                               category_guidance_scale=sc, hist_guidance_scale=sh, mutual_guidance_scale=sm, null_img=null_img,
                               eta=0.0, init_latents=None, generator=torch.Generator().manual_seed(SEED), output_type="latent",
                               return_dict=True)
    final, init_drawn = out[0].images, out[-1]
    fill = torch.nonzero(olists == 0)
    fill_cate = cats[fill[:, 0], fill[:, 1]]
    hist_sel = torch.stack([hist[int(uids[o])][int(c)] if int(c) in hist[int(uids[o])] else null_img
                            for (o, _), c in zip(fill.tolist(), fill_cate)])
    prompts = m.text_encoder(ids[fill[:, 0], fill[:, 1]])[0]
    null_prompt = m.text_encoder(torch.zeros(1, 77, dtype=torch.long))[0]
    mg.npsave(f"sample_{name}.npz", olists=olists, all_latents=images.reshape(bsz * 4, 4, H, H), null_latent=null_img,
              init_latents=init_drawn, hist_sel=hist_sel, category_prompts=prompts, null_prompt=null_prompt,
              scales=np.array([sc, sh, sm]), steps=steps, sched="ddim", use_history=True, use_mutual=True,
              timesteps=torch.stack([c["t"] for c in m.unet.calls]), n_calls=len(m.unet.calls), final=final,
              generator_seed=SEED, unet_checksum=mg.checksum(mg.tiny_weights()),
              **{f"{key}_{tag}": m.unet.calls[i][src] if key != "ehs_rows" else m.unet.calls[i]["ehs"][:, 0, :4]
                 for tag, i in (("0", 0), ("1", 1), ("last", len(m.unet.calls) - 1))
                 for key, src in (("x_in", "x"), ("ehs_rows", "ehs"), ("unet_out", "out"))},
              **{f"enc.{k}": v for k, v in mg.enc_state(m).items()})
    print("init drawn", tuple(init_drawn.shape), "final", tuple(final.shape), "calls", len(m.unet.calls))


if __name__ == "__main__":
    main()
