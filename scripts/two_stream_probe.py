#!/usr/bin/env python3
"""Would two half-batch U-Net forwards on two HIP streams overlap the memory-bound launches (GroupNorm / LayerNorm / short-K linears)
of one with the MFMA-bound launches of the other?  One batch-16 forward against two concurrent batch-8 forwards (two contexts with
the same weights), and against the two batch-8 forwards run back to back on one stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import difashion_amd as da

dev = "cuda"


def make(max_batch, sd=None):
    u = da.UNet2DConditionModel(sample_size=64, in_channels=8, max_batch=max_batch, init_seed=0).to(dev).eval()
    if sd is not None:
        u.load_state_dict(sd)
    u.pack()
    return u


u16 = make(16)
sd = u16.state_dict()
ua, ub = make(8, sd), make(8, sd)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(16, 8, 64, 64, device=dev, generator=g)
t = torch.randint(0, 1000, (16,), device=dev, generator=g)
e = torch.randn(16, 77, 768, device=dev, generator=g)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def one16():
    with torch.no_grad():
        u16(x, t, e)


def two8_serial():
    with torch.no_grad():
        ua(x[:8], t[:8], e[:8]); ub(x[8:], t[8:], e[8:])


def two8_streams():
    with torch.no_grad():
        with torch.cuda.stream(s1):
            ua(x[:8], t[:8], e[:8])
        with torch.cuda.stream(s2):
            ub(x[8:], t[8:], e[8:])


print(f"one batch-16 forward            : {timed(one16):7.2f} ms")
print(f"two batch-8 forwards, one stream : {timed(two8_serial):7.2f} ms")
print(f"two batch-8 forwards, two streams: {timed(two8_streams):7.2f} ms")
