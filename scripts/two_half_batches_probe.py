#!/usr/bin/env python3
"""Would two half batches of the sampling forward on two streams beat one batch of 16?  Two U-Net objects (same synthetic weights, own
workspaces) each run B = 8 on their own stream; the pair is timed against one B = 16 forward.  (Probe: the product runs one batch.)
   python scripts/two_half_batches_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = torch.device("cuda", 0)
ua, _ = bench.build_models(dev, "sd15")
ub, _ = bench.build_models(dev, "sd15")
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(16, 8, 64, 64, device=dev, generator=g)
e = torch.randn(16, 77, 768, device=dev, generator=g)
t = torch.full((16,), 481.0, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def one():
    with torch.no_grad():
        ua(x, t, e, return_dict=False)


def two():
    with torch.no_grad():
        with torch.cuda.stream(sa):
            ua(x[:8], t[:8], e[:8], return_dict=False)
        with torch.cuda.stream(sb):
            ub(x[8:], t[8:], e[8:], return_dict=False)


def seq():
    with torch.no_grad():
        ua(x[:8], t[:8], e[:8], return_dict=False)
        ub(x[8:], t[8:], e[8:], return_dict=False)


for name, fn in (("one batch of 16", one), ("two batches of 8, two streams", two), ("two batches of 8, one stream", seq), ("one batch of 16", one)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print(f"{name:34s} {(time.perf_counter() - t0) / 20 * 1e3:7.2f} ms per 16 rows", flush=True)
