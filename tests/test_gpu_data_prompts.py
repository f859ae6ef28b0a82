"""-m gpu legs of SURVEY.md 8f-2 / 8f-4: the prompt cache and the batch builder on the device path.

  * f2: ``PromptTable`` built ON the GPU and gathered ON the GPU equals the reference's per-batch ``text_encoder(ids)[0]`` calls
    (DiFashion/models/difashion.py:218-224 training, :340-353 sampling) -- and feeds ``DiFashion.fashion_generation`` in place of
    the text encoder with an identical result.
  * f4: ``preprocess_dataset`` (data_utils.py:87-161) with the item latents produced by THIS package's HIP ``AutoencoderKL``
    (no ``all_item_latents.npy`` cache) against ``oracle/vae_ref.py`` latents put through the same bookkeeping: the history
    condition is the mean of the history items' latents, ``hist["null"]`` the latent of item 0; a second call must read the
    cache it wrote and return the same tensors.
Tolerance: the VAE's stated one (bf16 path vs fp32 oracle, relative L2 <= 3e-2); everything after the latents is exact."""
import os
import types

import numpy as np
import pytest
import torch

import difashion_amd as da
from difashion_amd.data import FashionDiffusionData, category_prompt, preprocess_dataset
from difashion_amd.prompts import PromptTable
from oracle import vae_ref
from tests.gpu_util import DEV, rel_err
from tests.helpers_data import StubTokenizer, synthetic_dataset
from tests.test_gpu_vae import hip_vae

pytestmark = pytest.mark.gpu


class DeviceTextEncoder(torch.nn.Module):
    """Stand-in for the frozen CLIP text model, on the GPU: embedding + position table + a per-row nonlinearity (no GEMM: a
    batch-size dependent GEMM algorithm would make "encode once" differ from "encode per batch" in the last bit)."""

    def __init__(self, dim=64, vocab=1001, length=16):
        super().__init__()
        g = torch.Generator().manual_seed(21)
        self.emb = torch.nn.Parameter(torch.randn(vocab, dim, generator=g))
        self.pos = torch.nn.Parameter(torch.randn(length, dim, generator=g) * 0.1)
        self.mix = torch.nn.Parameter(torch.randn(dim, generator=g))

    def forward(self, ids):
        h = self.emb[ids] + self.pos[None, :ids.shape[1]]
        return (torch.tanh(h * self.mix) + h.mean(dim=1, keepdim=True),)


def test_prompt_table_on_device_equals_per_batch_encoding():
    _, id_cate, _, _ = synthetic_dataset()
    enc, tok = DeviceTextEncoder().to(DEV), StubTokenizer()
    table = PromptTable.build(enc, tok, id_cate, DEV, batch_size=3)
    assert table.table.device.type == "cuda" and table.table.shape == (len(id_cate) + 1, 16, 64)
    cats = torch.tensor([[1, 2, 3, 5], [4, 6, 1, 3], [5, 5, 2, 4]], device=DEV)          # ids already on the device: no host sync
    got = table.lookup(cats)
    ids = StubTokenizer()([category_prompt(id_cate[int(c)]) for c in cats.reshape(-1).cpu()], max_length=16, padding="max_length",
                          truncation=True, return_tensors="pt").input_ids
    with torch.no_grad():
        want = enc(ids.to(DEV))[0]
        null = enc(StubTokenizer()([""], max_length=16, padding="max_length", truncation=True, return_tensors="pt").input_ids.to(DEV))[0]
    assert got.device.type == "cuda" and torch.equal(got, want)
    assert torch.equal(table.null_prompt, null)
    # CPU ids work too (they are moved), the order is the row-major order of the ids
    assert torch.equal(table.lookup(cats.cpu()[1]), want[4:8])


def test_preprocess_dataset_through_the_hip_vae(tmp_path):
    cfg = vae_ref.TINY_VAE
    params = vae_ref.init_params(cfg, seed=2, w_std=0.05, affine_jitter=0.1)
    vae = hip_vae(cfg, params)
    data, id_cate, history, _ = synthetic_dataset()
    g = torch.Generator().manual_seed(17)
    images = torch.rand(13, 3, cfg.sample_size, cfg.sample_size, generator=g) * 2 - 1      # item id -> image tensor
    with torch.no_grad():
        ref_lat = vae_ref.encode(params, cfg, images) * cfg.scaling_factor
    out, hist = preprocess_dataset({k: list(v) for k, v in data.items()}, str(tmp_path), id_cate, history, images, StubTokenizer(), vae, DEV)
    cache = os.path.join(str(tmp_path), "all_item_latents.npy")
    assert os.path.exists(cache)
    lat = torch.tensor(np.load(cache, allow_pickle=True))
    e = rel_err(lat, ref_lat)
    print(f"item latents through the HIP VAE: rel err {e:.2e}")
    assert lat.shape == ref_lat.shape and e <= 3e-2
    # history bookkeeping on top of those latents is exact arithmetic on whatever the VAE produced ...
    for uid, per_cate in history.items():
        for cate, iids in per_cate.items():
            assert torch.equal(hist[uid][cate], lat[iids].mean(dim=0))
            assert rel_err(hist[uid][cate], ref_lat[iids].mean(dim=0)) <= 3e-2              # ... and within tolerance of the oracle's
    assert torch.equal(hist["null"], lat[0])
    assert [t.tolist() for t in out["outfits"]] == data["outfits"] and out["input_ids"][0].shape == (4, 16)
    # second call: the cache is read, the VAE is not needed
    out2, hist2 = preprocess_dataset({k: list(v) for k, v in data.items()}, str(tmp_path), id_cate, history, None, StubTokenizer(), None, DEV)
    assert torch.equal(hist2[7][5], hist[7][5]) and torch.equal(hist2["null"], hist["null"])
    batch = FashionDiffusionData(out2)[1]
    assert batch["uids"] == 7 and batch["category"].tolist() == [4, 6, 1, 3]
