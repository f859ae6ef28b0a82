"""Batch builder either side of the path: what DiFashion/data_utils.py does between the ``.npy`` files of the
dataset and the tensors the training / sampling glue consumes.

Mirrors (same names, arguments and results):
  * ``preprocess_dataset`` (data_utils.py:87-161): category ids -> text prompts -> token ids; item images -> VAE latents
    (``all_item_latents.npy`` cache, ``latent_dist.mode() * scaling_factor``, batches of 64) -- on THIS package's HIP
    AutoencoderKL when the cache is missing; per-user per-category history condition = mean of the history items' latents
    (+ the ``"null"`` entry = latent of item 0);
  * ``FashionDiffusionData`` / ``FashionFITBData`` (data_utils.py:47-84): index wrappers the DataLoader iterates.
Host-side bookkeeping is plain Python / torch indexing (plumbing); the only compute is the VAE encode.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Iterable, List, Sequence

import numpy as np
import torch
from torch.utils.data import Dataset

PAIR_CATEGORIES = ("pants", "earrings")      # categories phrased as "a pair of ..." (data_utils.py:103-107)


def category_prompt(category: str) -> str:
    article = "a pair of " if any(s in category for s in PAIR_CATEGORIES) else "a "
    return "A photo of " + article + category + ", on white background, high quality"


def tokenize_categories(outfit_categories: Iterable[Sequence[int]], id_cate_dict: Dict[int, str], tokenizer) -> List[torch.Tensor]:
    """One (olen, model_max_length) id tensor per outfit, padded / truncated like the reference's tokenizer call."""
    out = []
    for cids in outfit_categories:
        prompts = [category_prompt(id_cate_dict[c]) for c in cids]
        enc = tokenizer(prompts, max_length=tokenizer.model_max_length, padding="max_length", truncation=True, return_tensors="pt")
        out.append(enc.input_ids)
    return out


@torch.no_grad()
def encode_item_latents(vae, img_dataset, device, batch_size: int = 64) -> torch.Tensor:
    """Every item image -> scaled latent mode, in batches (the expensive step of preprocessing; HIP VAE)."""
    vae = vae.to(device)
    chunks = []
    n = len(img_dataset)
    for start in range(0, n, batch_size):
        imgs = torch.stack([img_dataset[i] for i in range(start, min(start + batch_size, n))], dim=0)
        imgs = imgs.to(memory_format=torch.contiguous_format).float().to(device)
        chunks.append(vae.encode(imgs).latent_dist.mode() * vae.config.scaling_factor)
    return torch.cat(chunks, dim=0).cpu()


def history_latents(all_latents: torch.Tensor, history: Dict) -> Dict:
    """hist[uid][category] = mean latent of that user's history items of the category; hist["null"] = latent of item 0."""
    hist: Dict = {}
    for uid, per_cate in history.items():
        slot = hist.setdefault(uid, {})
        for cate, iids in per_cate.items():
            slot[cate] = all_latents[iids].mean(dim=0)
    hist["null"] = all_latents[0]
    return hist


def preprocess_dataset(data, data_path, id_cate_dict, history, img_dataset, tokenizer, vae, device):
    data["input_ids"] = tokenize_categories(data["category"], id_cate_dict, tokenizer)
    cache = os.path.join(data_path, "all_item_latents.npy")
    if os.path.exists(cache):
        all_latents = torch.tensor(np.load(cache, allow_pickle=True))
    else:
        all_latents = encode_item_latents(vae, img_dataset, device)
        np.save(cache, np.array(all_latents))
    hist = history_latents(all_latents, history)
    data["category"] = [torch.tensor(c) for c in data["category"]]
    data["outfits"] = [torch.tensor(o).long() for o in data["outfits"]]
    return data, hist


class FashionDiffusionData(Dataset):
    KEYS = ("uids", "oids", "outfits", "input_ids", "category")

    def __init__(self, data):
        self.data = data

    def __len__(self):
        return len(self.data["uids"])

    def __getitem__(self, index):
        return {k: self.data[k][index] for k in self.KEYS}


class FashionFITBData(Dataset):
    """Fill-in-the-blank: the ground-truth outfit with its first ``fill_num`` slots zeroed (= to be generated)."""

    def __init__(self, data, all_test_grd, fill_num: int = 1):
        self.data, self.test_grd, self.fill_num = data, all_test_grd, fill_num

    def __len__(self):
        return len(self.data["uids"])

    def __getitem__(self, index):
        outfits = torch.tensor(self.test_grd["outfits"][index])
        outfits[:self.fill_num] = 0
        item = {k: self.data[k][index] for k in ("uids", "oids", "input_ids", "category")}
        item["outfits"] = outfits
        return item
