#!/usr/bin/env python3
"""Diagnosis: batch-row consistency and oracle error of the tiny U-Net with the LayerNorm fold on / off (DFH_LN_FOLD, read once per process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import difashion_amd as da
from difashion_amd import _lib
from oracle import unet_ref
from tests.gpu_util import DEV, rel_err
from tests.test_gpu_unet import hip_unet, inputs

cfg = unet_ref.TINY
params = unet_ref.init_params(cfg, seed=5)
m = hip_unet(cfg, params)
x, e = inputs(cfg, 3, 13)
t = torch.tensor([5, 300, 900])
with torch.no_grad():
    ref = unet_ref.unet_forward(params, cfg, x, t, e)
    xd, ed, td = x.to(DEV), e.to(DEV), t.to(DEV)
    _lib.census_reset()
    full = m(xd, td, ed).sample
    c3 = _lib.census()
    taps3 = {k: m.debug_tap(k).cpu() for k in ("conv_in", "down0", "down1", "down2", "mid", "up1", "up3")}
    for i in range(3):
        _lib.census_reset()
        one = m(xd[i:i + 1], td[i:i + 1], ed[i:i + 1]).sample
        c1 = _lib.census()
        taps1 = {k: m.debug_tap(k).cpu() for k in taps3}
        print(f"fold={os.environ.get('DFH_LN_FOLD', '1')} row {i}: B=3 vs B=1 {rel_err(full[i:i+1], one):.2e}  vs oracle: B=3 {rel_err(full[i:i+1].cpu(), ref[i:i+1]):.2e} "
              f"B=1 {rel_err(one.cpu(), ref[i:i+1]):.2e}", {k: f"{rel_err(taps3[k][i:i+1], taps1[k]):.1e}" for k in taps3})
    print("census B=3", {k: v for k, v in c3.items() if v})
    print("census B=1", {k: v for k, v in c1.items() if v})
    again = m(xd, td, ed).sample
    print("deterministic:", bool(torch.equal(again, full)))
