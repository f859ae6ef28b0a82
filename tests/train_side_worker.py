"""Worker of tests/test_gpu_train.py::test_side_stream_weight_gradients_are_bit_identical_to_the_one_stream_walk: one forward + backward
of a small U-Net in train mode, every gradient saved to argv[1].  The environment (DFH_TRAIN_SIDE, DFH_TRAIN_SIDE_MIN_FLOP) is read once
per process by the library, hence a process per setting."""
import os
import sys
import time

_T0 = time.time()


def _stamp(what):
    if os.environ.get("DFH_WORKER_TIMING"):
        print(f"[worker {time.time() - _T0:7.2f}s] {what}", file=sys.stderr, flush=True)


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import unet_ref
from tests.gpu_util import DEV
from tests.test_gpu_train import LINEAR_CFG
from tests.test_gpu_unet import hip_unet, inputs


def main():
    _stamp("imports done")
    cfg = LINEAR_CFG
    params = unet_ref.init_params(cfg, seed=5)
    m = hip_unet(cfg, params, max_batch=6).train()
    _stamp("model on device")
    x, e = inputs(cfg, 6, 77)
    t = torch.tensor([3, 250, 500, 750, 990, 41], device=DEV)
    grads = {}
    for rep in range(2):                      # twice: the second backward runs with warm caches and other timing
        for p in m.parameters():
            p.grad = None
        out = m(x.to(DEV), t, e.to(DEV)).sample
        g = torch.Generator().manual_seed(1)
        out.backward(torch.randn(out.shape, generator=g).to(DEV) / 64)
        torch.cuda.synchronize()
        grads[rep] = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
        _stamp(f"rep {rep} done")
    torch.save(grads, sys.argv[1])
    _stamp("saved")


if __name__ == "__main__":
    main()
