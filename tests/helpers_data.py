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
