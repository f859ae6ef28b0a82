#!/usr/bin/env python3
"""fp32-oracle outputs of the full-size U-Net parity legs of tests/test_gpu_unet.py, computed ONCE here (build container, CPU) so the
GPU box only compares (tests/oracle_cache.py explains the format; VERDICT r05 item 6: the GPU suite was at 695 s of a 1200 s limit).
Every case makes the same ``oracle.unet_ref.unet_forward`` call its test would make live; seeds, shapes and timesteps below mirror the tests.

    python tests/golden/make_golden_unet_full.py [case ...]        # sd15_b1 sd15_b16 sd15_b64_rows sd2base_b1
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import unet_ref  # noqa: E402
from tests import oracle_cache  # noqa: E402


def inputs(cfg, B, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cfg.in_channels, cfg.sample_size, cfg.sample_size, generator=g)
    e = torch.randn(B, 77, cfg.cross_attention_dim, generator=g)
    return x, e


def case(name):
    if name == "sd15_b1":
        cfg = unet_ref.SD15
        x, e = inputs(cfg, 1, 123)
        return cfg, x, torch.tensor([481]), e, None, ("conv_in", "down0", "down2", "mid", "up1", "up3")
    if name == "sd15_b16":
        cfg = unet_ref.SD15
        x, e = inputs(cfg, 16, 123)
        t = torch.tensor([981, 981, 981, 981, 741, 741, 741, 741, 501, 501, 501, 501, 21, 21, 21, 21])
        return cfg, x, t, e, None, ("conv_in", "down0", "down1", "down2", "mid", "up1", "up2", "up3")
    if name == "sd15_b64_rows":
        cfg = unet_ref.SD15
        x, e = inputs(cfg, 64, 777)
        t = torch.tensor([981, 741, 501, 21]).repeat_interleave(16)
        return cfg, x, t, e, [0, 21, 42, 63], ()
    if name == "sd2base_b1":
        cfg = unet_ref.SD2BASE
        x, _ = inputs(cfg, 1, 321)
        e = torch.randn(1, 77, cfg.cross_attention_dim, generator=torch.Generator().manual_seed(322))
        return cfg, x, torch.tensor([731]), e, None, ()
    raise SystemExit(f"unknown case {name}")


def main(names):
    torch.set_grad_enabled(False)
    cache = {}
    for name in names:
        cfg, x, t, e, rows, tap_keys = case(name)
        if cfg not in cache:
            cache.clear()
            cache[cfg] = unet_ref.init_params(cfg, seed=0)
        params = cache[cfg]
        t0 = time.time()
        taps = {}
        if rows is None:
            ref = unet_ref.unet_forward(params, cfg, x, t, e, taps=taps)
        else:
            ref = unet_ref.unet_forward(params, cfg, x[rows], t[rows], e[rows])
        rec = {"out": ref.numpy(), "fingerprint": oracle_cache.fingerprint(params, x, t, e)}
        for k in tap_keys:
            rec[f"tap_{k}"] = oracle_cache.subsample(taps[k])
        np.savez_compressed(os.path.join(HERE, f"unet_full_{name}.npz"), **rec)
        print(f"wrote unet_full_{name}.npz in {time.time() - t0:.1f} s: out {tuple(ref.shape)} |out| = {float(ref.norm()):.4f}, taps {list(tap_keys)}")


if __name__ == "__main__":
    main(sys.argv[1:] or ["sd15_b1", "sd15_b16", "sd15_b64_rows", "sd2base_b1"])
