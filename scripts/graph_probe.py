#!/usr/bin/env python3
"""Does replaying one sampling step from a hipGraph beat eager launches?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import difashion_amd as da

dev = torch.device("cuda", 0)
unet, enc = bench.build_models(dev, "sd15")
sampler = da.OutfitSampler(unet, enc, da.DDIMScheduler())
inp = bench.outfit_inputs(dev, 768, 0)
sampler.prepare(num_inference_steps=50, cate_scale=12.0, hist_scale=4.0, mutual_scale=5.0, eta=0.1, **inp)
for i in range(5):
    sampler.step(i)
torch.cuda.synchronize()

def timed(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

print("eager  ms/step:", timed(lambda: sampler.step(10)))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        sampler.step(10)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        sampler.step(10)
    print("graph  ms/step:", timed(g.replay))
except Exception as e:
    print("capture failed:", repr(e)[:500])
