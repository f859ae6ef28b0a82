#!/usr/bin/env python3
"""When do the ranges of the packed gradient arena become final during the SD-1.5 backward (batch 32)?  Prints, per range
handed out by dfh_unet_backward_next, the time on the compute stream since the start of the backward."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import difashion_amd as da
from difashion_amd import _lib

dev = torch.device("cuda")
B = 32
unet = da.UNet2DConditionModel(sample_size=64, in_channels=8, max_batch=B, init_seed=0).to(dev).train()
x = torch.randn(B, 8, 64, 64, device=dev); e = torch.randn(B, 77, 768, device=dev); t = torch.randint(0, 1000, (B,), device=dev)
for rep in range(2):
    out = unet(x, t, e).sample
    dout = torch.randn_like(out).contiguous().float()
    plist = unet.grad_views()
    arr = (C.c_void_p * len(plist))(*[p.grad.data_ptr() for p in plist])
    sp = _lib.stream_ptr()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True); ev0.record()
    _lib.call_count("dfh_unet_backward_begin", unet._ctx, _lib.ptr(dout), None, (256 << 20) // 4, sp)
    lo, hi, rec = C.c_size_t(0), C.c_size_t(0), []
    while True:
        rc = _lib.raw().dfh_unet_backward_next(unet._ctx, C.byref(lo), C.byref(hi), sp)
        assert rc >= 0
        if rc == 0: break
        ev = torch.cuda.Event(enable_timing=True); ev.record(); rec.append((lo.value, hi.value, ev))
    evn = torch.cuda.Event(enable_timing=True); evn.record()
    _lib.call("dfh_unet_backward_finish", unet._ctx, arr, len(plist), 1, sp)
    torch.cuda.synchronize()
total = unet._train_buffers[1].numel() // 4
done = 0
for l, h, ev in rec:
    done += h - l
    print(f"range [{l / 2**20:8.1f}, {h / 2**20:8.1f}) Mfloat  ({(h - l) * 4 / 2**20:6.0f} MB) final at {ev0.elapsed_time(ev):6.1f} ms   cumulative {100.0 * done / total:5.1f} %")
print(f"backward walk {ev0.elapsed_time(evn):.1f} ms; arena {total * 4 / 2**30:.2f} GiB")
