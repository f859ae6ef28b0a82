#!/usr/bin/env python3
"""Golden vectors for the batch builder from the REAL reference function.  Build container only
(needs /root/reference); the committed ``data_prep.npz`` is what travels.

``DiFashion/data_utils.py`` dies at import here (torchvision is absent): a ``sys.modules`` stub supplies that NAME only
(no arithmetic).  ``preprocess_dataset`` then runs on a synthetic dataset with an existing ``all_item_latents.npy`` (so it
takes its cache branch and needs no VAE) and a deterministic stand-in tokenizer that records the prompts it is given.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
sys.modules["torchvision"].transforms = types.ModuleType("torchvision.transforms")
sys.modules.setdefault("torchvision.transforms", sys.modules["torchvision"].transforms)
sys.path.insert(0, "/root/reference/DiFashion")
import data_utils as ref  # noqa: E402

sys.path.insert(0, os.path.dirname(HERE))
from helpers_data import StubTokenizer, synthetic_dataset  # noqa: E402


def main():
    data, id_cate, history, latents = synthetic_dataset()
    tok = StubTokenizer()
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "all_item_latents.npy"), latents.numpy())
        out, hist = ref.preprocess_dataset(data, d, id_cate, history, None, tok, None, "cpu")
    rec = {"n_outfits": np.array(len(out["input_ids"]))}
    for i, ids in enumerate(out["input_ids"]):
        rec[f"input_ids_{i}"] = ids.numpy()
        rec[f"category_{i}"] = out["category"][i].numpy()
        rec[f"outfits_{i}"] = out["outfits"][i].numpy()
    rec["prompts"] = np.array(tok.seen)
    rec["hist_null"] = hist["null"].numpy()
    for uid in history:
        for cate in history[uid]:
            rec[f"hist_{uid}_{cate}"] = hist[uid][cate].numpy()
    # FITB wrapper
    fitb = ref.FashionFITBData(out, {"outfits": [o.tolist() for o in out["outfits"]]}, fill_num=2)[1]
    rec["fitb_outfits_1"] = fitb["outfits"].numpy()
    np.savez_compressed(os.path.join(HERE, "data_prep.npz"), **rec)
    print("wrote data_prep.npz with", len(rec), "arrays")


if __name__ == "__main__":
    main()
