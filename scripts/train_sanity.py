#!/usr/bin/env python3
"""Full-size (SD-1.5 shape) training sanity: 40 optimisation steps on ONE fixed synthetic batch -- the loss must fall and stay finite."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import difashion_amd as da

dev = torch.device("cuda", 0)
unet, enc = bench.build_models(dev, "sd15")
unet.train(); enc.train()
opt = da.FusedAdamW(list(unet.parameters()) + list(enc.parameters()), lr=5e-5, weight_decay=1e-2, max_grad_norm=1.0)
ema = da.EMAModel(unet.parameters())
sched = da.DDIMScheduler()
kw = bench.train_inputs(dev, 768, 0, 8)
losses, norms = [], []
for i in range(40):
    losses.append(da.train_step(unet, enc, sched, opt, ema_unet=ema, **kw))
    norms.append(opt.grad_norm().clone())
losses = [float(x) for x in losses]; norms = [float(x) for x in norms]
print("loss :", " ".join(f"{v:.4f}" for v in losses[::4]))
print("gnorm:", " ".join(f"{v:.3f}" for v in norms[::4]))
assert all(v == v and v < 10 for v in losses) and losses[-1] < 0.7 * losses[0], "training diverged or stalled"
print("ok: loss", losses[0], "->", losses[-1])
