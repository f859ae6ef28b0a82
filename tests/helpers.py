"""Shared test helpers (oracle side).  Tests may import oracle/; the product may not."""
import os

import numpy as np
import torch

from oracle import unet_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# the tiny U-Net the golden glue fixtures were captured with (tests/golden/make_golden.py)
GLUE_CFG = unet_ref.UNetConfig(sample_size=16, block_out_channels=(32, 64, 128, 128),
                               cross_attention_dim=64, num_heads=(1, 2, 2, 2))


def glue_unet_params():
    return unet_ref.init_params(GLUE_CFG, seed=7, w_std=0.05, affine_jitter=0.1)


def checksum(params):
    return np.array([float(sum(v.double().sum() for v in params.values())),
                     float(sum((v.double() ** 2).sum() for v in params.values()))])


def load(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    out = {}
    for k in z.files:
        v = z[k]
        out[k] = torch.from_numpy(np.asarray(v)) if v.dtype.kind in "fiub" else v
    return out


def enc_params(rec):
    return {k[4:]: v for k, v in rec.items() if k.startswith("enc.")}


def rel_err(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))
