#!/usr/bin/env python3
"""How long does the HOST take to enqueue one training step (no sync inside) vs the device time of the step?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import difashion_amd as da

dev = torch.device("cuda", 0)
unet, enc = bench.build_models(dev, "sd15")
unet.train(); enc.train()
opt = da.FusedAdamW(list(unet.parameters()) + list(enc.parameters()), lr=1e-5, max_grad_norm=1.0)
sched = da.DDIMScheduler()
kw = bench.train_inputs(dev, 768, 0, 8)
for _ in range(2):
    da.train_step(unet, enc, sched, opt, **kw)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    loss = da.train_forward(unet, enc, sched, **kw)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step(); opt.zero_grad()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"enqueue: forward {1e3*(t1-t0):.1f} ms, backward {1e3*(t2-t1):.1f} ms, optimizer {1e3*(t3-t2):.1f} ms; device drained after {1e3*(t4-t0):.1f} ms")
