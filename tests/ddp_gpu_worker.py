"""Worker of tests/test_gpu_ddp.py: launched as 2 ranks by torch.distributed.run, BOTH on cuda:0 (the GPU box has one device),
process group on gloo (which all-reduces CUDA tensors through the host; RCCL refuses two ranks on one device).  Everything
above the collective is the product's data-parallel training step as it runs under RCCL: native backward into the flat
gradient buffer, ONE all-reduce of that buffer, the fused clip + AdamW + EMA on every rank.

Checks: (1) the averaged gradient of two half-batches equals the gradient of the whole batch computed
by one process (same kernels, other summation order: RTOL), (2) the loss of the whole batch is the mean of the shard losses,
(3) after the optimizer step the replicas hold bit-identical parameters and EMA shadows.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as tdist

import difashion_amd as da
from difashion_amd import dist as ddist
from difashion_amd.pipeline import train_forward
from tests.helpers import GLUE_CFG, glue_unet_params, load
from tests.test_gpu_train import batch_kwargs, make_encoder
from tests.test_gpu_unet import hip_unet

RTOL = 2e-2
# RCCL ("nccl") with one rank per device when the box has at least two; otherwise both ranks share cuda:0 over gloo
# (torch.cuda.device_count() does not initialise the GPU)
NCCL = torch.cuda.device_count() >= 2 and os.environ.get("DFH_DIST_BACKEND", "nccl") == "nccl"
DEV = f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}" if NCCL else "cuda:0"


def shard(kw, rank, world, olen=4):
    """Outfits [rank * b/world, (rank+1) * b/world) of the batch: per-outfit tensors by outfit, per-item tensors by 4 items."""
    b = kw["timesteps_outfit"].shape[0]
    lo, hi = rank * b // world, (rank + 1) * b // world
    out = dict(kw)
    out["timesteps_outfit"] = kw["timesteps_outfit"][lo:hi]
    for k in ("latents", "noise", "hist_latents", "ehs", "random_p", "random_p_cate", "dropout_mask"):
        if kw.get(k) is not None:
            out[k] = kw[k][lo * olen:hi * olen]
    return out


def build(rec):
    unet = hip_unet(GLUE_CFG, glue_unet_params(), max_batch=32).train()
    enc = make_encoder(rec)
    opt = da.FusedAdamW(list(unet.parameters()) + list(enc.parameters()), lr=2e-4, weight_decay=1e-2, max_grad_norm=1.0)
    ddist.broadcast_parameters(opt.flat_param)        # the unused category embedding is initialised from each process's RNG
    ema = da.EMAModel(unet.parameters(), decay=0.9999)
    return unet, enc, opt, ema


def main():
    rank, world, _ = ddist.init("nccl" if NCCL else "gloo")
    assert world == 2 and tdist.get_backend() == ("nccl" if NCCL else "gloo")
    torch.cuda.set_device(torch.device(DEV))
    print(f"[rank {rank}] backend {tdist.get_backend()} device {DEV}", flush=True)
    rec = load("train_b8_snr5.npz")
    sched = da.DDIMScheduler(prediction_type=str(rec["pred_type"]))
    kw = batch_kwargs(rec, DEV)

    # whole batch, one process, no collective: the reference gradient
    unet, enc, opt, _ = build(rec)
    loss_full = train_forward(unet, enc, sched, **kw)
    loss_full.backward()
    g_full = opt.flat_grad.clone()
    del unet, enc, opt

    # the product's data-parallel step on this rank's half of the outfits: (a) gradients averaged INSIDE backward, range by
    # range on a side stream while the walk continues (the default; small ranges here so the tiny U-Net has many), (b) one
    # all-reduce of the flat buffer after backward
    mine = shard(kw, rank, world)
    # the bf16 wire format (dist.exchange_bf16: all_to_all + fp32 accumulate + all_gather) against the same whole-batch gradient
    unet, enc, opt, ema = build(rec)
    unet.grad_wire_dtype, unet.grad_bucket_bytes, unet.measure_comm = "bf16", 64 << 10, True
    da.train_step(unet, enc, sched, opt, ema_unet=ema, **mine)
    err16 = float((opt.flat_grad - g_full).norm() / g_full.norm())
    exposed = unet.last_comm_exposed_ms()
    print(f"[rank {rank}] bf16 wire: grad rel err {err16:.3e}; compute stream waited {exposed:.3f} ms for the exchange", flush=True)
    assert err16 < RTOL and exposed is not None and exposed >= 0.0

    def digest(t):
        return t.view(torch.int32).to(torch.int64).sum().item() if t.dtype == torch.float32 else 0
    d16 = torch.tensor([digest(opt.flat_param), digest(ema.flat), digest(opt.flat_grad)], dtype=torch.int64, device=DEV if NCCL else "cpu")
    both16 = [torch.zeros_like(d16) for _ in range(world)]
    tdist.all_gather(both16, d16)
    assert torch.equal(both16[0], both16[1]), both16          # identical averages on every rank -> identical replicas
    # the exchange allocates nothing on the device once its buffers exist (dist.Bf16Exchange): the second step's overlapped backward
    # must leave the caching allocator's allocation count where it was
    orig = unet._backward_overlapped
    counts = {}

    def counted(*a, **k):
        torch.cuda.synchronize()
        before = torch.cuda.memory_stats()["allocation.all.allocated"]
        out = orig(*a, **k)
        torch.cuda.synchronize()
        counts["delta"] = torch.cuda.memory_stats()["allocation.all.allocated"] - before
        return out

    unet._backward_overlapped = counted
    da.train_step(unet, enc, sched, opt, ema_unet=ema, **mine)
    unet._backward_overlapped = orig
    print(f"[rank {rank}] device allocations inside the overlapped backward (second step): {counts.get('delta')}", flush=True)
    assert counts.get("delta") == 0, counts
    del unet, enc, opt, ema
    for overlapped in (False, True):
        unet, enc, opt, ema = build(rec)
        unet.sync_grads_in_backward = overlapped
        unet.grad_bucket_bytes = 64 << 10
        loss = da.train_step(unet, enc, sched, opt, ema_unet=ema, **mine)
        assert unet.grads_synced == overlapped
        g_dp = opt.flat_grad                   # averaged gradient; the lazy zero_grad keeps the buffer
        err = float((g_dp - g_full).norm() / g_full.norm())
        mean_loss = ddist.sum_over_ranks(float(loss)) / world
        print(f"[rank {rank}] overlapped={overlapped} shard loss {float(loss):.5f} mean {mean_loss:.5f} full {float(loss_full):.5f}  "
              f"grad rel err {err:.3e}", flush=True)
        assert err < RTOL, err
        assert abs(mean_loss - float(loss_full)) < 2e-3 * abs(float(loss_full)) + 1e-5
        if not overlapped:
            g_flat = g_dp.clone()
            del unet, enc, opt, ema
    err2 = float((g_dp - g_flat).norm() / g_flat.norm())      # the two exchange schemes against each other
    print(f"[rank {rank}] overlapped vs flat all-reduce: {err2:.3e}", flush=True)
    assert err2 < 1e-5, err2

    # replicas stay bit-identical: compare a 64-bit digest of parameters and EMA shadows across the ranks
    mine_d = torch.tensor([digest(opt.flat_param), digest(ema.flat)], dtype=torch.int64, device=DEV if NCCL else "cpu")
    both = [torch.zeros_like(mine_d) for _ in range(world)]
    tdist.all_gather(both, mine_d)
    assert torch.equal(both[0], both[1]), both
    # a second step keeps them identical and still reduces ONE flat buffer
    da.train_step(unet, enc, sched, opt, ema_unet=ema, **mine)
    mine_d = torch.tensor([digest(opt.flat_param), digest(ema.flat)], dtype=torch.int64, device=DEV if NCCL else "cpu")
    tdist.all_gather(both, mine_d)
    assert torch.equal(both[0], both[1]), both
    # freshness is collective: a parameter whose gradient was written on ONE rank only (a data-dependent branch) holds
    # the same averaged gradient everywhere after the exchange and must be updated everywhere, as torch DDP would -- here the category
    # embedding, which nothing uses in the product's loss: stamped on rank 0 only
    emb = enc.category_embedding.weight
    opt.zero_grad(lazy_modules=[unet])
    emb.grad.fill_(0.25)
    if rank == 0:
        opt.mark_fresh([emb])
    before = emb.detach().clone()
    opt.step()
    moved = float((emb.detach() - before).abs().max())
    mine_d = torch.tensor([digest(opt.flat_param), 0], dtype=torch.int64, device=DEV if NCCL else "cpu")
    tdist.all_gather(both, mine_d)
    print(f"[rank {rank}] embedding stamped on rank 0 only: moved by {moved:.3e} on this rank", flush=True)
    assert moved > 0.0 and torch.equal(both[0], both[1]), (moved, both)
    torch.cuda.synchronize()
    tdist.barrier()
    if rank == 0:
        print("DDP_GPU_OK", flush=True)


if __name__ == "__main__":
    main()
