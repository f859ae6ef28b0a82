#!/usr/bin/env python3
"""Launch census of ONE sampling step of the bench workload (which kernel families the walk takes at U-Net batch 16 / 64): GPU only.
    python scripts/census_step.py [outfits]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import difashion_amd as da
from difashion_amd import _lib
dev = torch.device("cuda", 0)
outfits = int(sys.argv[1]) if len(sys.argv) > 1 else 1
unet, enc = bench.build_models(dev, "sd15")
s = da.OutfitSampler(unet, enc, da.DDIMScheduler())
s.prepare(num_inference_steps=50, cate_scale=12.0, hist_scale=4.0, mutual_scale=5.0, eta=0.1, **bench.outfit_inputs(dev, 768, 0, outfits))
s.step(0); s.step(1)
torch.cuda.synchronize()
_lib.census_reset()
s.step(2)
torch.cuda.synchronize()
print(f"U-Net batch {16 * outfits}:", {k: v for k, v in _lib.census().items() if v})
