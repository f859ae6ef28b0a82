"""Worker of tests/test_gpu_ddp.py::test_rccl_branch_runs_in_a_world_of_one: ONE rank, backend "nccl" (= RCCL), DFH_DIST_SINGLE_RANK=1.
A one-GPU box cannot run two RCCL ranks (RCCL refuses two ranks on one device), so the 2-rank test runs over gloo, which stages through
the host and blocks -- it cannot show a stream-ordering mistake around a collective.  Here every collective of the data-parallel training
step really goes through RCCL on its side stream (all_reduce / all_to_all_single / all_gather_into_tensor / broadcast), enqueued
asynchronously behind events exactly as with eight ranks; with one rank the average is the identity, so the results are known:

  1. exchange_bf16(t) == t rounded to bf16 (pack, all_to_all, one-contribution mean, all_gather, unpack), bit for bit;
  2. all_reduce_gradients(fp32) and broadcast_parameters leave their buffers bit-identical;
  3. two optimisation steps with the gradients exchanged INSIDE the backward (ranges on the side stream, fp32 wire) end with the
     gradients, parameters and EMA shadows of two steps without any collective (to the last bits of the float atomics of the bias
     gradients); with the bf16 wire the gradients are the bf16 roundings of those;
  4. the second step allocates nothing on the device inside the overlapped backward."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as tdist

import difashion_amd as da
from difashion_amd import dist as ddist
from tests.helpers import GLUE_CFG, glue_unet_params, load
from tests.test_gpu_train import batch_kwargs, make_encoder
from tests.test_gpu_unet import hip_unet

DEV = "cuda:0"


def build(rec):
    torch.manual_seed(0)
    unet = hip_unet(GLUE_CFG, glue_unet_params(), max_batch=32).train()
    enc = make_encoder(rec)
    opt = da.FusedAdamW(list(unet.parameters()) + list(enc.parameters()), lr=2e-4, weight_decay=1e-2, max_grad_norm=1.0)
    ddist.broadcast_parameters(opt.flat_param)
    ema = da.EMAModel(unet.parameters(), decay=0.9999)
    return unet, enc, opt, ema


def glue_start(rec):
    """The flat parameter buffer before any step (same construction as build)."""
    os.environ["DFH_DIST_SINGLE_RANK"] = "0"
    _, _, opt, _ = build(rec)
    return opt.flat_param.clone()


def two_steps(rec, sched, kw, collectives, wire):
    os.environ["DFH_DIST_SINGLE_RANK"] = "1" if collectives else "0"
    unet, enc, opt, ema = build(rec)
    unet.grad_wire_dtype, unet.grad_bucket_bytes, unet.measure_comm = wire, 64 << 10, collectives
    grads = []
    for step in range(2):
        if step == 1 and collectives:
            orig, counts = unet._backward_overlapped, {}

            def counted(*a, **k):
                torch.cuda.synchronize()
                before = torch.cuda.memory_stats()["allocation.all.allocated"]
                out = orig(*a, **k)
                torch.cuda.synchronize()
                counts["delta"] = torch.cuda.memory_stats()["allocation.all.allocated"] - before
                return out
            unet._backward_overlapped = counted
        da.train_step(unet, enc, sched, opt, ema_unet=ema, **kw)
        assert unet.grads_synced == collectives
        grads.append(opt.flat_grad.clone())
    torch.cuda.synchronize()
    if collectives:
        assert counts.get("delta") == 0, counts
        assert unet.last_comm_exposed_ms() is not None and getattr(unet, "_comm_stream", None) is not None
    return opt.flat_param.clone(), ema.flat.clone(), grads


def main():
    torch.cuda.set_device(0)
    tdist.init_process_group("nccl", rank=0, world_size=1)
    assert tdist.get_backend() == "nccl" and tdist.get_world_size() == 1
    os.environ["DFH_DIST_SINGLE_RANK"] = "1"
    assert ddist.active()
    g = torch.Generator(device=DEV).manual_seed(3)
    for n in (1000003, 17, 4096):                                # ragged and tiny lengths: padding of the wire shards
        t = torch.randn(n, device=DEV, generator=g)
        want = t.to(torch.bfloat16).float()
        got = ddist.exchange_bf16(t.clone())
        torch.cuda.synchronize()
        assert torch.equal(got, want), (n, float((got - want).abs().max()))
    t = torch.randn(300001, device=DEV, generator=g)
    assert torch.equal(ddist.all_reduce_gradients(t.clone(), wire="fp32"), t)
    assert torch.equal(ddist.broadcast_parameters(t.clone()), t)
    assert float(ddist.gather_mean(torch.tensor(2.5, device=DEV))) == 2.5
    print("collectives of a one-rank RCCL world: identities hold", flush=True)

    rec = load("train_b8_snr5.npz")
    sched = da.DDIMScheduler(prediction_type=str(rec["pred_type"]))
    kw = batch_kwargs(rec, DEV)
    p0, e0, g0 = two_steps(rec, sched, kw, collectives=False, wire="fp32")
    p1, e1, g1 = two_steps(rec, sched, kw, collectives=True, wire="fp32")

    def rel(a, b):
        return float((a - b).norm() / b.norm())
    # same kernels, same order; bias / norm-scale gradients are accumulated with float atomics in either run, hence not bit for bit
    r = [rel(g1[0], g0[0]), rel(g1[1], g0[1]), rel(p1 - p0 + p0, p0), rel(e1, e0)]
    print("fp32 wire vs no collective: gradient step 1 / step 2 / parameters / EMA relative differences " + " ".join(f"{v:.2e}" for v in r), flush=True)
    # (measured 2e-8 .. 4e-8 / 1.5e-5 .. 4.2e-4 / 4e-7 .. 1.6e-5 / 3e-7 .. 1.3e-5 over repeated runs: the atomics' last bits of step 1 move AdamW's
    # normalised update of near-zero gradients and flip a few bf16 roundings of the re-packed weights in step 2)
    assert r[0] < 1e-6 and r[1] < 5e-3 and r[2] < 2e-4 and r[3] < 2e-4, r
    p2, e2, g2 = two_steps(rec, sched, kw, collectives=True, wire="bf16")
    # the U-Net's ranges went over the wire as bf16, the encoder's through the flat exchange as bf16 too: every gradient = bf16 of the exact one
    r16 = rel(g2[0], g0[0].to(torch.bfloat16).float())
    upd = float((p2 - p1).norm() / (p1 - glue_start(rec)).norm())
    print(f"bf16 wire: gradients vs bf16 of the fp32-wire gradients {r16:.2e}; parameter UPDATE after two steps differs by {upd:.2e} (relative)", flush=True)
    # (AdamW divides by the running RMS of the gradient: a 2^-9 rounding of the gradients moves the two-step update by a few per cent; 3.2e-2 measured)
    assert r16 < 1e-4 and upd < 1e-1, (r16, upd)
    print("RCCL single-rank run OK", flush=True)
    tdist.destroy_process_group()


if __name__ == "__main__":
    main()
