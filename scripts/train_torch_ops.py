"""Which torch-level ops (copies, fills, random draws) one training step of bench.py --mode train issues beside the native walk.
Usage (GPU box): python scripts/train_torch_ops.py > gpurun_out/train_torch_ops.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import difashion_amd as da
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
unet, enc = bench.build_models(dev, "sd15")
unet.train(); enc.train()
opt = da.FusedAdamW(list(unet.parameters()) + list(enc.parameters()), lr=1e-5, weight_decay=1e-2, max_grad_norm=1.0)
ema = da.EMAModel(unet.parameters())
sched = da.DDIMScheduler()
kw = bench.train_inputs(dev, unet.config.cross_attention_dim, 0, 8)
step = lambda: da.train_step(unet, enc, sched, opt, ema_unet=ema, **kw)
for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=40, max_name_column_width=70))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="count", row_limit=25, max_name_column_width=60, max_src_column_width=110))
