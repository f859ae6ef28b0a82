#!/usr/bin/env python3
"""VAE encode at the training batch (32 images of 512x512, difashion.py:144) and decode of one outfit: time + per-class table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import difashion_amd as da
from difashion_amd import _lib

dev = "cuda"
vae = da.AutoencoderKL(init_seed=None).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for n, p in vae.named_parameters():
        if n.endswith(".weight") and "norm" not in n.split(".")[-2]:
            p.normal_(0.0, 0.02, generator=g)
for B, fn_name in ((32, "encode"), (4, "encode"), (4, "decode")):
    x = torch.rand(B, 3, 512, 512, device=dev) * 2 - 1 if fn_name == "encode" else torch.randn(B, 4, 64, 64, device=dev)
    fn = (lambda: vae.encode(x)) if fn_name == "encode" else (lambda: vae.decode(x))
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    _lib.prof_begin()
    fn()
    cl = _lib.prof_end()
    print(f"{fn_name} B={B}: {ms:.2f} ms  ({B / ms * 1e3:.0f} images/s)  " +
          "  ".join(f"{c}: {v['ms']:.2f} ms" + (f" {v['flops'] / v['ms'] / 1e9:.0f} TF" if v['flops'] else "") for c, v in cl.items() if v["launches"]))
