"""Case table shared by tests/golden/make_golden_clip.py (writes the fixtures from the REAL transformers class), tests/test_clip_cpu.py
(oracle vs fixtures) and tests/test_gpu_clip.py (HIP encoder vs fixtures).  Weights and token ids are regenerated from the seeds by
``oracle.clip_ref.init_params`` / ``prompt_like_ids`` (CPU generator: the same tensors on every box); the fixtures hold the real
class's outputs plus a checksum of the inputs they were computed from."""
import os

import numpy as np
import torch

from oracle import clip_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> (config, weight seed, batch, sequence length, keep all hidden states in the fixture)
CASES = {
    "tiny_quickgelu": (clip_ref.TINY_CLIP, 11, 5, 77, True),
    "tiny_short_seq": (clip_ref.TINY_CLIP, 12, 3, 20, True),
    "tiny_gelu_eos": (clip_ref.TINY_CLIP_GELU, 13, 4, 77, True),
    "sd15_clip_l": (clip_ref.SD15_CLIP, 14, 3, 77, False),          # CLIP ViT-L/14 text tower: what SD-1.5's text_encoder/ holds
    "sd2_openclip_h": (clip_ref.SD2_CLIP, 15, 2, 77, False),        # OpenCLIP ViT-H/14, 23 layers: SD-2-base, the reference's default
}


def case_inputs(name):
    cfg, seed, batch, T, _ = CASES[name]
    return cfg, clip_ref.init_params(cfg, seed), clip_ref.prompt_like_ids(cfg, batch, T, seed + 1000)


def checksum(params, ids) -> np.ndarray:
    """Order-independent fingerprint of the regenerated inputs (float64 sums of a few tensors + the ids)."""
    keys = sorted(params)
    pick = [keys[0], keys[len(keys) // 2], keys[-1]]
    return np.array([float(params[k].double().sum()) for k in pick] + [float(params[k].double().abs().sum()) for k in pick]
                    + [float(ids.double().sum())])


def load_fixture(name):
    return dict(np.load(os.path.join(GOLDEN, f"clip_{name}.npz")))


def rel(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.double() - b.double()).norm() / b.double().norm())
