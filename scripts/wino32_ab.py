#!/usr/bin/env python3
"""A/B asked for by the round-3 review: Winograd F(2x2,3x3) also at the 32x32 level (640-channel resnet convs), measured instead of
extrapolated -- output error of the batch-16 SD-1.5 forward against the fp32 oracle's output is not recomputed here (minutes of CPU);
the script reports the difference between the two HIP walks and, with --oracle, both against the oracle.
    DFH_WINO_MAXHW=1024 python scripts/wino32_ab.py [--oracle]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import unet_ref
from tests.test_gpu_unet import hip_unet, inputs, DEV, rel_err
from difashion_amd import _lib
cfg = unet_ref.SD15
params = unet_ref.init_params(cfg, seed=0)
x, e = inputs(cfg, 16, 123)
t = torch.tensor([981] * 4 + [741] * 4 + [501] * 4 + [21] * 4)
m = hip_unet(cfg, params, max_batch=16)
with torch.no_grad():
    m(x.to(DEV), t.to(DEV), e.to(DEV)); torch.cuda.synchronize()
    _lib.census_reset()
    out = m(x.to(DEV), t.to(DEV), e.to(DEV)).sample.cpu()
    print("conv_wino launches:", _lib.census()["conv_wino"], "DFH_WINO_MAXHW =", os.environ.get("DFH_WINO_MAXHW", "256"))
torch.save(out, f"gpurun_out/wino_out_{os.environ.get('DFH_WINO_MAXHW', '256')}.pt")
if "--oracle" in sys.argv:
    with torch.no_grad():
        ref = unet_ref.unet_forward(params, cfg, x, t, e)
    torch.save(ref, "gpurun_out/wino_ref.pt")
    print("rel err vs fp32 oracle:", rel_err(out, ref))
elif os.path.exists("gpurun_out/wino_ref.pt"):
    print("rel err vs fp32 oracle:", rel_err(out, torch.load("gpurun_out/wino_ref.pt")))
