"""Multi-GPU launch plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl").

Sampling shards over OUTFITS with no data-path collective: the four items of an
outfit are coupled every step through the mutual condition (difashion.py:475-490) so an outfit stays
on one GPU, while different outfits are independent.  The only collectives of the sampling path are control-plane: a
barrier around the timed region, a MAX over ranks of the elapsed time, and (optionally, outside the
timed region) an all_gather of the finished latents.  Training shards the batch of outfits the same
way and has ONE real exchange step per optimizer step: the gradient all-reduce (all_reduce_gradients).
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the launcher environment (torchrun / torch.distributed.run)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init(backend: str | None = None) -> Tuple[int, int, int]:
    """Initialise the default process group when WORLD_SIZE > 1.  backend: "nccl" (= RCCL on ROCm) on
    GPUs, "gloo" for the CPU tests."""
    rank, world, local = env_world()
    single = os.environ.get("DFH_DIST_SINGLE_RANK") == "1"      # a world of one rank that still runs its collectives (see active())
    if (world > 1 or single) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def active(group=None) -> bool:
    """Do the gradient collectives run?  A process group with more than one rank -- or, with ``DFH_DIST_SINGLE_RANK=1``, any initialised
    group: a world of ONE rank then still goes through every RCCL call of the data-parallel step (all_reduce, all_to_all_single,
    all_gather_into_tensor, broadcast on the side stream, with the event ordering around them), which is how the ``nccl`` branch is
    exercised on a one-GPU box (tests/test_gpu_ddp.py::test_rccl_branch_runs_in_a_world_of_one)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("DFH_DIST_SINGLE_RANK") == "1"


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous shard of ``n_items`` outfits for ``rank``: sizes differ by at most one, earlier ranks
    take the remainder (every outfit assigned exactly once, order preserved)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def _reduce_device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float) -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_reduce_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float) -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_reduce_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_mean(value: torch.Tensor) -> torch.Tensor:
    """Mean over the ranks of a per-rank tensor (the logged loss: ``accelerator.gather(loss.repeat(b)).mean()``,
    train.py:695, with equal per-rank batches).  Returns a new tensor; nothing on the optimisation path depends on it."""
    out = value.detach().clone().float()
    if active():
        if dist.get_backend() != "nccl":
            out = out.cpu()
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
        out = out / dist.get_world_size()
    return out


class Bf16Exchange:
    """Persistent buffers of the bf16-wire gradient exchange: ``wire`` (this rank's range rounded to bf16, later the gathered averages),
    ``recv`` (the W contributions to this rank's shard) and ``shard`` -- sized once for the largest range (``capacity`` floats) and reused
    for every range of every step, so the exchange allocates NOTHING on the device inside the backward (2 x 128 MB per
    256 MB range, ~14 ranges per step before).  One object per (device, capacity); ``for_range`` hands out the cached one."""

    _cache = {}

    def __init__(self, capacity: int, device, world: int):
        self.world = world
        self.per_max = ((capacity + world - 1) // world + 7) // 8 * 8          # shard length: a multiple of 8 (16-byte kernels)
        self.wire = torch.zeros(world * self.per_max, dtype=torch.bfloat16, device=device)
        self.recv = torch.zeros(world * self.per_max, dtype=torch.bfloat16, device=device)
        self.shard = torch.zeros(self.per_max, dtype=torch.bfloat16, device=device)

    @classmethod
    def for_range(cls, n: int, device, world: int, capacity: int = 0) -> "Bf16Exchange":
        # one set of buffers per (device, world, STREAM): the exchange inside the backward runs on a side stream, the flat exchange of
        # the non-overlapped path on the compute stream -- buffers shared across streams could be replaced (and handed back to the
        # allocating stream's pool) while the other stream still reads them
        key = (str(device), world, torch.cuda.current_stream(device).cuda_stream)
        ex = cls._cache.get(key)
        need = max(n, capacity)
        if ex is None or ex.per_max * world < ((need + world - 1) // world + 7) // 8 * 8 * world:
            if ex is not None:
                torch.cuda.current_stream(device).synchronize()      # the old buffers may still be in flight on this stream
            ex = cls(need, device, world)
            cls._cache[key] = ex
        return ex

    @classmethod
    def release(cls) -> None:
        """Drop every cached buffer set (after training, or before a phase that needs the memory)."""
        if cls._cache and torch.cuda.is_available():
            torch.cuda.synchronize()
        cls._cache.clear()


# largest range (floats) one pass of the bf16 exchange handles: longer ranges (the flat gradient buffer of the non-overlapped path:
# 870 M floats) go through the same persistent buffers chunk by chunk instead of pinning 2 x n bf16 for the life of the process
BF16_EXCHANGE_MAX = 64 << 20


def exchange_bf16(t: torch.Tensor, capacity: int = 0) -> torch.Tensor:
    """Gradient average with a bf16 WIRE format and fp32 accumulation, in place on the fp32 range ``t``: half the bytes of an
    fp32 all-reduce on every xGMI link.  Direct reduce-scatter + all-gather, the natural pattern of the fully connected 8-GPU
    mesh (each pair of GPUs owns a link):
      1. every rank rounds its range to bf16 (dfh_wire_pack) and sends shard j to rank j (all_to_all: (W-1)/W of the bf16 range leaves
         the GPU);
      2. rank j sums the W bf16 contributions of its shard IN FP32, in rank order (deterministic), divides by W, rounds to bf16 -- ONE
         kernel (dfh_wire_shard_mean);
      3. the averaged shards are all-gathered in bf16 ((W-1)/W of the bf16 range arrives) and widened back to fp32 (dfh_wire_unpack).
    Every rank ends with the SAME values (each shard is computed by exactly one rank), so the replicas stay bit-identical.
    The buffers are persistent (Bf16Exchange; ``capacity`` = the largest range the caller will pass, in floats): no device allocation
    per call.  gloo has no all_to_all: there step 1 is an all_gather of the whole bf16 range staged through the host (test path only,
    same arithmetic, same device buffers)."""
    if not active():
        return t
    world, rank = dist.get_world_size(), dist.get_rank()
    n = t.numel()
    flat = t.reshape(-1)
    if t.is_cuda and n > BF16_EXCHANGE_MAX:
        for lo in range(0, n, BF16_EXCHANGE_MAX):
            exchange_bf16(flat[lo:lo + BF16_EXCHANGE_MAX], capacity=BF16_EXCHANGE_MAX)
        return t
    if not t.is_cuda:                       # CPU tensors (gloo unit tests): the same arithmetic in torch ops
        per = (n + world - 1) // world
        wire = torch.zeros(world * per, dtype=torch.bfloat16)
        wire[:n].copy_(flat)
        stage = wire.view(torch.uint8)      # raw bytes: gloo transports neither bf16 nor int16
        bufs = [torch.empty_like(stage) for _ in range(world)]
        dist.all_gather(bufs, stage)
        mine = torch.stack([b[2 * rank * per:2 * (rank + 1) * per] for b in bufs]).view(torch.bfloat16)
        acc = mine[0].float()
        for r in range(1, world):
            acc += mine[r].float()
        shard = (acc / world).to(torch.bfloat16)
        bufs = [torch.empty(2 * per, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(bufs, shard.view(torch.uint8))
        flat.copy_(torch.cat(bufs).view(torch.bfloat16)[:n])
        return t
    from . import _lib
    ex = Bf16Exchange.for_range(n, t.device, world, capacity)
    per = ((n + world - 1) // world + 7) // 8 * 8
    wire, recv, shard = ex.wire[:world * per], ex.recv[:world * per], ex.shard[:per]
    sp = _lib.stream_ptr()
    _lib.call("dfh_wire_pack", _lib.ptr(flat), _lib.ptr(wire), n, world * per, sp)
    if dist.get_backend() == "nccl":
        dist.all_to_all_single(recv, wire)                       # recv[r * per:(r + 1) * per] = rank r's copy of MY shard
    else:
        torch.cuda.current_stream(t.device).synchronize()
        stage = wire.cpu().view(torch.uint8)
        bufs = [torch.empty_like(stage) for _ in range(world)]
        dist.all_gather(bufs, stage)
        recv.copy_(torch.cat([b[2 * rank * per:2 * (rank + 1) * per] for b in bufs]).view(torch.bfloat16))
    _lib.call("dfh_wire_shard_mean", _lib.ptr(recv), _lib.ptr(shard), world, per, sp)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(wire, shard)
    else:
        torch.cuda.current_stream(t.device).synchronize()
        stage = shard.cpu().view(torch.uint8)
        bufs = [torch.empty_like(stage) for _ in range(world)]
        dist.all_gather(bufs, stage)
        wire.copy_(torch.cat(bufs).view(torch.bfloat16))
    _lib.call("dfh_wire_unpack", _lib.ptr(wire), _lib.ptr(flat), n, sp)
    return t


def all_reduce_gradients(flat_grad: torch.Tensor, average: bool = True, wire: str = "fp32") -> torch.Tensor:
    """Data-parallel gradient exchange of the training step (the reference: accelerate's DDP wrapper around
    accelerator.backward, train.py:611/:699).  Every rank holds a full replica and a different slice of the global
    batch of outfits; the gradients of ALL parameters live in one flat fp32 buffer (training.FusedAdamW.flat_grad /
    UNet2DConditionModel.grad_views), so the exchange is ONE RCCL all-reduce sized for xGMI instead of DDP's
    25 MB buckets: a ring over 8 GPUs moves 2 * 7/8 of the buffer per link once, with no per-bucket launch latency."""
    if not active():
        return flat_grad
    if wire == "bf16":
        if not average:
            raise ValueError("the bf16 wire format averages")
        return exchange_bf16(flat_grad)
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    if average:
        flat_grad.div_(dist.get_world_size())
    return flat_grad


def broadcast_parameters(flat_param: torch.Tensor, src: int = 0) -> torch.Tensor:
    """Make every replica start from rank ``src``'s weights (what DDP does when it wraps a module, train.py:611): ONE
    broadcast of the flat fp32 parameter buffer (training.FusedAdamW.flat_param -- the parameters are views of it)."""
    if active():
        dist.broadcast(flat_param, src=src)
    return flat_param


def gather_outfit_latents(local: torch.Tensor, counts: Sequence[int]) -> torch.Tensor:
    """Concatenate every rank's finished latents in rank order (not on the timed path).
    ``counts[r]`` = number of latent rows rank r holds."""
    if not dist.is_initialized():
        return local
    world = dist.get_world_size()
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
