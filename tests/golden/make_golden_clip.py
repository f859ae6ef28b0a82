#!/usr/bin/env python3
"""Golden vectors for the CLIP text encoder from the REAL third-party class.  Build container only (needs the
installed ``transformers``; nothing of it travels to the GPU box): the committed ``clip_*.npz`` are what travels.

The reference calls ``transformers.CLIPTextModel`` (DiFashion/models/difashion.py:70-72, :224, :234, :340-342, :352).  For every case
of ``tests/helpers_clip.py`` this script builds that class from the case's config (eager attention, fp32, eval), loads the seeded
weights of ``oracle.clip_ref.init_params`` under the class's own state-dict names, runs it on the seeded prompt-like token ids
exactly as the reference does -- ``text_encoder(input_ids)`` with no attention mask -- and records ``last_hidden_state`` (= ``[0]``),
``pooler_output`` and the per-layer ``hidden_states`` (all of them for the tiny cases, the embeddings / middle / last layer for the
full-size ones), together with a checksum of the inputs.

    python tests/golden/make_golden_clip.py [case ...]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import transformers  # noqa: E402
from transformers import CLIPTextConfig, CLIPTextModel  # noqa: E402

from helpers_clip import CASES, case_inputs, checksum  # noqa: E402


def real_model(cfg, params):
    hf_cfg = CLIPTextConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                            num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                            max_position_embeddings=cfg.max_position_embeddings, hidden_act=cfg.hidden_act,
                            layer_norm_eps=cfg.layer_norm_eps, eos_token_id=cfg.eos_token_id, bos_token_id=cfg.bos_token_id,
                            pad_token_id=cfg.pad_token_id, attn_implementation="eager")
    model = CLIPTextModel(hf_cfg).eval().float()
    own = model.state_dict()
    # transformers 4.32.1 (the reference's pin) nests the tower under ``text_model.``; newer releases flattened it
    sd = {k: params[k if k in params else "text_model." + k] for k in own if not k.endswith("position_ids")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.endswith("position_ids") for m in missing), (missing, unexpected)
    assert len(sd) == len(params), (len(sd), len(params))
    return model


def main(names):
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    for name in names:
        cfg, params, ids = case_inputs(name)
        keep_all = CASES[name][4]
        model = real_model(cfg, params)
        out = model(ids, output_hidden_states=True)
        hs = out.hidden_states
        assert len(hs) == cfg.num_hidden_layers + 1
        taps = list(range(len(hs))) if keep_all else [0, cfg.num_hidden_layers // 2, cfg.num_hidden_layers]
        rec = {"input_ids": ids.numpy(), "last_hidden_state": out[0].numpy(), "pooler_output": out.pooler_output.numpy(),
               "taps": np.array(taps), "checksum": checksum(params, ids),
               "transformers_version": np.array(transformers.__version__)}
        for t in taps:
            rec[f"hidden_{t}"] = hs[t].numpy()
        np.savez_compressed(os.path.join(HERE, f"clip_{name}.npz"), **rec)
        print(f"wrote clip_{name}.npz: last_hidden_state {tuple(out[0].shape)}, |x| = {float(out[0].norm()):.4f}, taps {taps}")


if __name__ == "__main__":
    main(sys.argv[1:] or list(CASES))
