"""Synthetic iFashion-shaped dataset + stand-in tokenizer shared by tests/golden/make_golden_data.py and the CPU test."""
import zlib

import torch


class _Enc:
    def __init__(self, ids):
        self.input_ids = ids


class StubTokenizer:
    """Deterministic stand-in for the CLIP tokenizer: one id per word (crc32 of the word), padded to model_max_length."""
    model_max_length = 16

    def __init__(self):
        self.seen = []

    def __call__(self, prompts, max_length, padding, truncation, return_tensors):
        assert padding == "max_length" and truncation and return_tensors == "pt"
        rows = []
        for p in prompts:
            self.seen.append(p)
            ids = [zlib.crc32(w.encode()) % 1000 + 1 for w in p.replace(",", " ,").split()][:max_length]
            rows.append(ids + [0] * (max_length - len(ids)))
        return _Enc(torch.tensor(rows, dtype=torch.long))


def synthetic_dataset():
    id_cate = {0: "null", 1: "t-shirt", 2: "pants", 3: "sneakers", 4: "drop earrings", 5: "handbag", 6: "wide-leg pants"}
    data = {"uids": [7, 7, 9], "oids": [100, 101, 102],
            "outfits": [[3, 5, 8, 11], [2, 4, 6, 1], [9, 10, 7, 12]],
            "category": [[1, 2, 3, 5], [4, 6, 1, 3], [5, 5, 2, 4]]}
    history = {7: {1: [1, 3], 2: [5], 5: [2, 8, 11]}, 9: {3: [4, 6, 9, 12], 4: [10]}}
    g = torch.Generator().manual_seed(11)
    latents = torch.randn(13, 4, 8, 8, generator=g) * 0.18215
    return data, id_cate, history, latents


def evalio_case(root, task):
    """Synthetic inputs of the evaluation writer (inf4eval.py:774): 14 ground-truth item JPEGs on disk, two batches of
    ``fashion_generation(..., return_dict=False)``-shaped results (PIL images), the second repeating an (uid, oid) of the first."""
    import os

    import numpy as np
    from PIL import Image
    rng = np.random.RandomState(5)
    img_root = os.path.join(root, "items")
    os.makedirs(img_root)
    paths = []
    for i in range(14):
        paths.append(f"item_{i}.jpg")
        Image.fromarray(rng.randint(0, 255, (12, 10, 3), dtype=np.uint8)).save(os.path.join(img_root, paths[-1]))
    grd = {100: {"outfits": [3, 5, 8, 11]}, 101: {"outfits": [2, 4, 6, 1]}, 102: {"outfits": [9, 10, 7, 12]}}
    cates = {100: [1, 2, 3, 5], 101: [4, 6, 1, 3], 102: [5, 6, 2, 4]}

    def pil(n):
        return [Image.fromarray(rng.randint(0, 255, (16, 16, 3), dtype=np.uint8)) for _ in range(n)]

    def rec(oid, nfill):
        full = torch.tensor(cates[oid])
        outfit = torch.tensor(grd[oid]["outfits"])
        outfit[:nfill] = 0
        return {"images": pil(nfill), "cates": [full[i] for i in range(nfill)], "full_cates": full, "outfits": outfit}

    nfill = 4 if task == "GOR" else 2
    batches = [{7: {100: rec(100, nfill), 101: rec(101, nfill if task == "GOR" else 1)}},
               {9: {102: rec(102, nfill)}, 7: {100: rec(100, nfill)}}]
    return dict(batches=batches, gen=os.path.join(root, f"{task}-checkpoint-10-cate12.0-mutual5.0-hist4.0"), img_root=img_root,
                paths=paths, grd=grd)
