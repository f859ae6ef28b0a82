#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): U-Net denoise steps/sec for a 4-item outfit at 64x64x4 latents.

  python bench.py --gpus N --steps K --warmup W
  N > 1: either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...: RANK /
  WORLD_SIZE come from the environment) or plainly as `python bench.py --gpus N`: the process then starts N ranks of itself
  (launch_ranks), one per GPU, and relays rank 0's line.

One "step" = everything one iteration of DiFashion.fashion_generation's loop does for one outfit
(DiFashion/models/difashion.py:456-577) at BASELINE.json configs[1]: sibling reduce + MutualEncoder MLP
+ input assembly + ONE U-Net forward at batch 16 (4 items x 4 guidance branches, SD-1.5 shape,
in_channels 8) + guidance combine + DDIM update.  Inputs are resident in HBM before the timed region.
Multi-GPU: one process per GPU, each denoising its own outfit (outfits are independent; no data-path
collective) -> weak scaling; value = N*K / max-over-ranks(elapsed).

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit/..., plus
  "roofline"     : dominant kernel (gemm_bf16_kernel: every conv3x3 / 1x1 / linear) -- ALGORITHMIC flops
                   per launch / average launch duration from HIP events recorded on the launch stream
                   (a second, profiled pass of the same K steps; the timed pass carries no events);
  "cpu_baseline" : the fp32 oracle (stand-in for the reference diffusers path) timed on
                   this box's host cores on a bounded sample;
  "secondary_configs" (default invocation, one GPU): short legs of the OTHER BASELINE configurations run after the headline's timed
                   region, so that one driver-run record witnesses them -- configs[4] (the same sampler with enable_fp8()), the
                   reference's real inference batch (4 outfits per call = U-Net batch 64, inf4eval.py:521-524) and configs[2] (the
                   training step); the headline fields and its timed region are untouched by them (--no-secondary skips them).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_BF16_PEAK = 2500.0   # TFLOP/s dense, MI355X_MICROARCH.md
MFMA_FP8_PEAK = 5000.0    # TFLOP/s dense, block-scaled e4m3 (MI355X_MICROARCH.md)
# What a register-resident loop of nothing but MFMAs sustains on this pool's boxes with RANDOM operands
# (scripts/probes/mfma_rate2.hip -> profiles/r02/mfma_rate2.txt: v_mfma_f32_32x32x16_bf16 issues every 32.0 cycles; 2.3-2.5 PFLOP/s
# at 2.3-2.4 GHz on zero-filled operands, 1.73-1.81 PFLOP/s on random ones because the clock drops to 1.75-1.95 GHz under dense
# MFMA load).  Informational only -- roofline.frac is against the nominal 2500.  (Round 1 quoted 1170 here: its probe's
# accumulator chains had been folded into one dependent chain by the compiler -- 45 cycles per MFMA instead of 16-20.)
MFMA_BF16_SUSTAINED = 1750.0
HBM_PEAK = 8000.0         # GB/s


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_models(dev, cfg_name):
    import difashion_amd as da
    kw = dict(sample_size=64, in_channels=8, max_batch=16, init_seed=None)
    if cfg_name == "sd2base":
        kw.update(cross_attention_dim=1024, attention_head_dim=(5, 10, 20, 20), use_linear_projection=True)
    t0 = time.time()
    unet = da.UNet2DConditionModel(**kw).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(0)
    with torch.no_grad():                      # synthetic weights: N(0, 0.02), norm gamma 1 / beta 0, biases 0
        for n, p in unet.named_parameters():
            is_norm = ".norm" in n or n.startswith("conv_norm_out")
            if n.endswith(".weight") and not is_norm:
                p.normal_(0.0, 0.02, generator=g)
            elif n.endswith(".weight"):
                p.fill_(1.0)
            else:
                p.zero_()
    enc = da.MutualEncoder(cate_num=50, cate_emb_size=64, latent_channels=4, latent_size=64, hid_dim=256).to(dev).eval()
    with torch.no_grad():
        for i in (0, 3):
            torch.nn.init.xavier_normal_(enc.mlp[i].weight)
            enc.mlp[i].bias.zero_()
    unet.pack()
    torch.cuda.synchronize()
    log(f"[bench] models ready in {time.time() - t0:.1f}s; workspace+arenas {unet.workspace_bytes() / 2**30:.2f} GiB")
    return unet, enc


def outfit_inputs(dev, cross_dim, rank, outfits=1):
    """Synthetic iFashion-shaped inputs: ``outfits`` 4-item outfits, every slot generated (GOR).  outfits = 4 is the
    reference's own inference batch (inf4eval.py:521-524: 4 outfits per fashion_generation call -> U-Net batch 64 under full CFG)."""
    n = 4 * outfits
    def rn(seed, *shape):
        return torch.randn(*shape, generator=torch.Generator().manual_seed(seed + 1000 * rank)).to(dev)
    return dict(olists=torch.zeros(outfits, 4, dtype=torch.long), all_latents=rn(122, n, 4, 64, 64) * 0.18215,
                init_latents=rn(123, n, 4, 64, 64), hist_latents=rn(124, n, 4, 64, 64) * 0.18215,
                null_latent=rn(125, 4, 64, 64) * 0.18215, category_prompts=rn(126, n, 77, cross_dim),
                null_prompt=rn(127, 1, 77, cross_dim))


def cpu_baseline(threads):
    """Oracle U-Net (fp32, torch CPU) + glue on the host cores, as SURVEY.md 8d / BASELINE.md 3 prescribe: 1 warm-up + 3 timed
    single forwards at B=1 (config 0) and ONE timed forward at B=16 (the rows of a step); steps/s = 1 / (t_fwd(B=16) + t_glue).
    The 50-step loop is never run on CPU."""
    from oracle import glue_ref, unet_ref
    torch.set_num_threads(threads)
    cfg = unet_ref.SD15
    p = unet_ref.init_params(cfg, seed=0)
    g = torch.Generator().manual_seed(123)
    x = torch.randn(16, 8, 64, 64, generator=g)
    e = torch.randn(16, 77, 768, generator=g)
    with torch.no_grad():
        unet_ref.unet_forward(p, cfg, x[:1], 481, e[:1])
        t1 = []
        for _ in range(3):
            t0 = time.perf_counter()
            unet_ref.unet_forward(p, cfg, x[:1], 481, e[:1])
            t1.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        unet_ref.unet_forward(p, cfg, x, 481, e)
        t16 = time.perf_counter() - t0
        enc = {"mlp.0.weight": torch.randn(256, 16384, generator=g) * 0.01, "mlp.0.bias": torch.zeros(256),
               "mlp.3.weight": torch.randn(16384, 256, generator=g) * 0.01, "mlp.3.bias": torch.zeros(16384)}
        lat = torch.randn(4, 4, 64, 64, generator=g)
        t0 = time.perf_counter()
        m = glue_ref.mutual_encoder(enc, glue_ref.mutual_sum(torch.zeros(1, 4, dtype=torch.long), lat, lat))
        xin = torch.cat([0.9 * torch.cat([lat] * 4) + 0.1 * torch.cat([m] * 4), torch.cat([lat] * 4)], 1)
        t_glue = time.perf_counter() - t0
    del p
    return dict(value=1.0 / (t16 + t_glue), unit="steps/s", cores=threads, kind="port",
                sample=f"oracle fp32 U-Net (SD-1.5 shape): 1 warm-up + 3 timed forwards at B=1 = {min(t1):.2f} / {sorted(t1)[1]:.2f} / {max(t1):.2f}s "
                       f"(min / median / max), 1 timed forward at B=16 = {t16:.2f}s, glue {t_glue * 1e3:.0f}ms; steps/s = 1 / (t_B16 + glue)")


def train_inputs(dev, cross_dim, rank, outfits):
    """Synthetic iFashion-shaped training batch (BASELINE configs[2]): ``outfits`` x 4 items, every tensor resident in HBM."""
    n = outfits * 4
    def rn(seed, *shape):
        return torch.randn(*shape, generator=torch.Generator().manual_seed(seed + 1000 * rank)).to(dev)
    g = torch.Generator().manual_seed(77 + rank)
    return dict(latents=rn(1, n, 4, 64, 64) * 0.18215, noise=rn(2, n, 4, 64, 64),
                timesteps_outfit=torch.randint(0, 1000, (outfits,), generator=g).to(dev), null_latent=rn(3, 4, 64, 64) * 0.18215,
                hist_latents=rn(4, n, 4, 64, 64) * 0.18215, ehs=rn(5, n, 77, cross_dim), null_prompt=rn(6, 1, 77, cross_dim),
                random_p=torch.rand(n, generator=g).to(dev), random_p_cate=torch.rand(n, generator=g).to(dev), snr_gamma=5.0,
                dropout_mask=((torch.rand(n, 256, generator=g) >= 0.1).float() / 0.9).to(dev))


def backend_name():
    import torch.distributed as tdist
    return tdist.get_backend() if tdist.is_initialized() else None


def min_over_ranks(ddist, v):
    return -ddist.max_over_ranks(-v)


# kernel classes of the training step against their roof (dfh_common.h ProfClass names): MFMA classes count the reference algorithm's
# multiply-adds (backward = data gradient + weight gradient of every forward GEMM, the attention backward its five products), the
# element-wise classes their algorithmic bytes
TRAIN_MFMA_CLASSES = ("gemm_conv3x3", "gemm_linear", "gemm_wgrad", "attention", "attention_bwd")
TRAIN_HBM_CLASSES = ("groupnorm", "layernorm", "norm_bwd", "splitk_reduce", "other", "optimizer")
TRAIN_CLASS_NOTE = {
    "gemm_conv3x3": "forward + data-gradient 3x3 / stride-2 / upsample convs (gemm_bf16_kernel / gemm_wide_kernel)",
    "gemm_linear": "forward + data-gradient 1x1 convs / linears / GEGLU (gemm_bf16_kernel / gemm_wide_kernel)",
    "gemm_wgrad": "weight gradients dW = dY^T X (gemm_wgrad_kernel + wgrad_reduce; large launches run on a side stream beside the data gradient)",
    "attention": "attention forward (attention_x32_kernel / attention_kernel)",
    "attention_bwd": "attention backward, P recomputed from the log-sum-exp (attention_bwd_kernel)",
    "optimizer": "clip + AdamW + EMA in one pass (adamw_kernel): 36 B per parameter",
}


def train_roofline(classes, K, elapsed):
    """Per-class roofline of the training step from the live HIP-event profile: MFMA classes against the dense bf16 peak, element-wise
    classes against HBM.  Durations are per launch on the launch's own stream (two streams overlap: their sum exceeds the wall time)."""
    per = {}
    for c, v in classes.items():
        if not v["launches"] or v["ms"] <= 0:
            continue
        if c in TRAIN_MFMA_CLASSES:
            tf = v["flops"] / (v["ms"] * 1e-3) / 1e12
            per[c] = dict(bound="mfma", achieved=round(tf, 1), peak=MFMA_BF16_PEAK, unit="TFLOP/s", frac=round(tf / MFMA_BF16_PEAK, 4),
                          ms_per_step=round(v["ms"] / K, 3), launches_per_step=v["launches"] // K, kernel=TRAIN_CLASS_NOTE.get(c, c))
        elif c in TRAIN_HBM_CLASSES:
            gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9
            per[c] = dict(bound="hbm", achieved=round(gb, 1), peak=HBM_PEAK, unit="GB/s", frac=round(gb / HBM_PEAK, 4),
                          ms_per_step=round(v["ms"] / K, 3), launches_per_step=v["launches"] // K, kernel=TRAIN_CLASS_NOTE.get(c, c))
    mf = sum(classes[c]["flops"] for c in TRAIN_MFMA_CLASSES if c in classes)
    mm = sum(classes[c]["ms"] for c in TRAIN_MFMA_CLASSES if c in classes)
    achieved = mf / (mm * 1e-3) / 1e12 if mm > 0 else 0.0
    return dict(bound="mfma", kernel="all MFMA-class launches of the step (forward + data-gradient GEMMs, weight-gradient GEMM, attention forward / backward)",
                achieved=round(achieved, 1), peak=MFMA_BF16_PEAK, unit="TFLOP/s", frac=round(achieved / MFMA_BF16_PEAK, 4), traffic=None,
                algorithmic_tflop_per_step=round(mf / K / 1e12, 2), mfma_class_ms_per_step=round(mm / K, 2),
                whole_step_frac=round(mf / K / (elapsed / K) / 1e12 / MFMA_BF16_PEAK, 4),
                note="achieved = summed algorithmic flops / summed launch durations of the MFMA classes (HIP events on each launch's stream; "
                     "the weight-gradient side stream overlaps the main stream, so the class times add up to more than ms_per_step); "
                     "whole_step_frac = the same flops over the wall-clock step",
                classes=per)


def cpu_baseline_train(threads):
    """The training step's CPU baseline: oracle fp32 U-Net forward + torch autograd backward of the MSE noise-prediction loss at B = 1
    (SD-1.5 shape, every parameter requires grad), one warm-up-free timed pass; items/s = 1 / t (a port: the reference trains through
    diffusers + autograd the same way).  Bounded: ~10-30 s of host work."""
    from oracle import unet_ref
    torch.set_num_threads(threads)
    cfg = unet_ref.SD15
    p = unet_ref.init_params(cfg, seed=0)
    for v in p.values():
        v.requires_grad_(True)
    g = torch.Generator().manual_seed(321)
    x = torch.randn(1, 8, 64, 64, generator=g)
    e = torch.randn(1, 77, 768, generator=g)
    noise = torch.randn(1, 4, 64, 64, generator=g)
    t0 = time.perf_counter()
    loss = torch.nn.functional.mse_loss(unet_ref.unet_forward(p, cfg, x, 481, e), noise)
    t_f = time.perf_counter() - t0
    loss.backward()
    t = time.perf_counter() - t0
    del p
    return dict(value=round(1.0 / t, 4), unit="items/s", cores=threads, kind="port",
                sample=f"oracle fp32 U-Net (SD-1.5 shape) forward {t_f:.2f}s + autograd backward {t - t_f:.2f}s of the MSE loss at B=1 "
                       "(no optimizer pass); items/s = 1 / (t_fwd + t_bwd)")


def measure_train(args, da, _lib, ddist, rank, world, dev, K, W, profile=True):
    """BASELINE configs[2]/[3]: one optimisation step = loss forward + native backward + RCCL gradient all-reduce + clip/AdamW
    + EMA over a per-GPU batch of 8 outfits x 4 items (weak scaling: global batch = 32 x N items).  Returns the result dict on
    rank 0 (None elsewhere)."""
    unet, enc = build_models(dev, args.config)
    unet.train(); enc.train()
    opt = da.FusedAdamW(list(unet.parameters()) + list(enc.parameters()), lr=1e-5, weight_decay=1e-2, max_grad_norm=1.0)
    ddist.broadcast_parameters(opt.flat_param)          # replicas start identical (DDP's construction-time broadcast)
    ema = da.EMAModel(unet.parameters())
    sched = da.DDIMScheduler()
    kw = train_inputs(dev, unet.config.cross_attention_dim, rank, args.outfits)
    # gradient wire: fp32 all-reduce per range unless --wire bf16 (the library default and the reference's DDP, train.py:611,699)
    wire = args.wire or "fp32"
    unet.grad_wire_dtype = wire
    unet.measure_comm = ddist.active()     # two event records per step on the compute stream around its wait for the side stream
    step = lambda: da.train_step(unet, enc, sched, opt, ema_unet=ema, **kw)
    for _ in range(W):
        loss = step()
    torch.cuda.synchronize(); ddist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    exposed = []
    for _ in range(K):
        loss = step()
        if ddist.active():
            exposed.append(unet._comm_events)          # read after the timed region (no host sync inside it)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0                     # this rank's own time, before it waits for the others
    ddist.barrier(); torch.cuda.synchronize()
    elapsed = ddist.max_over_ranks(time.perf_counter() - t0)
    own_max, own_min = ddist.max_over_ranks(own), min_over_ranks(ddist, own)
    comm_exposed_ms = comm_exposed_min = None
    if exposed and all(e is not None for e in exposed):
        mine = sum(e0.elapsed_time(e1) for e0, e1 in exposed) / len(exposed)
        comm_exposed_ms, comm_exposed_min = ddist.max_over_ranks(mine), min_over_ranks(ddist, mine)
    assert torch.isfinite(loss), "non-finite loss"
    classes = None
    if profile and rank == 0 and world == 1:
        _lib.prof_begin()
        for _ in range(K):
            step()
        dump = None
        if os.environ.get("DFH_PROF_TABLE"):                      # per-shape launch table of the training step (diagnosis)
            import tempfile
            dump = os.path.join(tempfile.gettempdir(), f"dfh_prof_dump_{os.getpid()}.txt")
            os.environ["DFH_PROF_DUMP"] = dump
        classes = _lib.prof_end()
        if dump:
            os.environ.pop("DFH_PROF_DUMP", None)
            per_launch_bound(dump, K)
    if world > 1:
        ddist.barrier()
    hbm = round(torch.cuda.max_memory_allocated() / 2**30, 1)
    single = world == 1 and ddist.active()
    bname = backend_name()
    del step, opt, ema, unet, enc, kw
    if rank != 0:
        return None
    items = args.outfits * 4
    out = {"metric": "training items/sec, fwd+bwd+AdamW, 8 outfits x 4 items per GPU @ 64x64x4 latent", "value": round(world * items * K / elapsed, 2),
           "unit": "items/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed * 1e3 / K, 2),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"BASELINE configs[2]: training step, {args.outfits} outfits x 4 items per GPU, {args.config} shape in_channels=8, "
                                  "min-SNR MSE loss, mutual + history conditioning, clip 1.0 + AdamW + EMA",
                      "unet_batch": items, "parallelism": f"data-parallel x{world}, gradients all-reduced over RCCL in 256 MB ranges of the packed arena, overlapped with the backward walk"},
           "loss": round(float(loss), 5), "hbm_GiB": hbm,
           # time per step the compute stream waited for the side-stream gradient exchange = the part of the exchange the backward
           # walk did not hide (max over ranks; null on one GPU: no exchange)
           "comm_exposed_ms": None if comm_exposed_ms is None else round(comm_exposed_ms, 3), "grad_wire": wire,
           "rccl_ranks": args.ranks_seen if args.backend in (None, "nccl") else None, "collective_backend": args.backend,
           "rccl_ranks_match_n_gpus": (args.ranks_seen == world) if args.backend in (None, "nccl") else None}
    if world > 1:       # a slow rank or an exposed exchange must be visible from this one record
        out["rank_ms_per_step"] = dict(min=round(own_min * 1e3 / K, 2), max=round(own_max * 1e3 / K, 2),
                                       note="each rank's own K steps up to its device sync, before the closing barrier")
        out["comm_exposed_ms_min"] = None if comm_exposed_min is None else round(comm_exposed_min, 3)
    if single:          # DFH_DIST_SINGLE_RANK=1: the gradient exchange ran over RCCL in a world of one rank
        out["single_rank_collectives"] = True
        out["collective_backend"] = bname
    out["roofline"] = None
    if classes is not None:
        tot_f = sum(v["flops"] for v in classes.values())
        out["roofline"] = train_roofline(classes, K, elapsed)
        out["kernel_classes"] = {c: dict(launches_per_step=v["launches"] // K, ms_per_step=round(v["ms"] / K, 3),
                                         tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] and v["ms"] > 0 else None,
                                         algorithmic_GBps=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None)
                                 for c, v in classes.items() if v["launches"]}
        out["algorithmic_tflop_per_step"] = round(tot_f / K / 1e12, 2)
        out["mfma_frac_whole_step"] = round(tot_f / K / (elapsed / K) / 1e12 / MFMA_BF16_PEAK, 4)
    return out


def run_train(args, da, _lib, ddist, rank, world, dev):
    out = measure_train(args, da, _lib, ddist, rank, world, dev, args.steps, args.warmup, profile=not args.no_profile)
    if out is None:
        return
    if not args.no_cpu_baseline and world == 1:
        try:
            out["cpu_baseline"] = cpu_baseline_train(min(16, len(os.sched_getaffinity(0))))
        except Exception as e:  # the GPU number must still be reported
            out["cpu_baseline"] = {"error": repr(e)}
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)


def vae_flops(boc=(128, 256, 512, 512), layers=2, latent=64):
    """Algorithmic FLOPs of one SD-VAE decode / encode (2 x MACs of every conv / linear / attention product)."""
    def conv(cin, cout, hw, k=9):
        return 2.0 * k * cin * cout * hw
    def resnet(cin, cout, hw):
        return conv(cin, cout, hw) + conv(cout, cout, hw) + (conv(cin, cout, hw, 1) if cin != cout else 0.0)
    def mid(c, hw):
        return 2 * resnet(c, c, hw) + 4 * conv(c, c, hw, 1) + 2 * 2.0 * hw * hw * c
    rev, hw = tuple(reversed(boc)), latent * latent
    dec = conv(4, 4, hw, 1) + conv(4, rev[0], hw) + mid(rev[0], hw)
    ch = rev[0]
    for i, oc in enumerate(rev):
        for j in range(layers + 1):
            dec += resnet(ch if j == 0 else oc, oc, hw)
        ch = oc
        if i != len(rev) - 1:
            hw *= 4
            dec += conv(ch, ch, hw)
    dec += conv(boc[0], 3, hw)
    enc = conv(3, boc[0], hw)
    ch = boc[0]
    for i, oc in enumerate(boc):
        for j in range(layers):
            enc += resnet(ch if j == 0 else oc, oc, hw)
        ch = oc
        if i != len(boc) - 1:
            hw //= 4
            enc += conv(ch, ch, hw)
    enc += mid(ch, hw) + conv(ch, 8, hw) + conv(8, 8, hw, 1)
    return dec, enc


def run_vae(args, da, _lib, ddist, rank, world, dev):
    """the VAE on either side of the denoising path -- decode of one outfit's four 64x64x4 latents to
    512x512 images (difashion.py:580) and encode of four 512x512 images (:144); random SD-VAE-shape weights."""
    vae = da.AutoencoderKL(init_seed=None).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(0)
    with torch.no_grad():
        for n, p in vae.named_parameters():
            if n.endswith(".weight") and "norm" not in n.split(".")[-2]:
                p.normal_(0.0, 0.02, generator=g)
    z = torch.randn(4, 4, 64, 64, device=dev)
    x = torch.rand(4, 3, 512, 512, device=dev) * 2 - 1
    K, W = args.steps, args.warmup
    res = {}
    dec_f, enc_f = vae_flops()
    for name, fn, fl in (("decode", lambda: vae.decode(z), dec_f), ("encode", lambda: vae.encode(x), enc_f)):
        for _ in range(W):
            fn()
        torch.cuda.synchronize(); ddist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        torch.cuda.synchronize(); ddist.barrier(); torch.cuda.synchronize()
        el = ddist.max_over_ranks(time.perf_counter() - t0)
        res[name] = dict(ms_per_outfit=round(el * 1e3 / K, 2), images_per_s=round(world * 4 * K / el, 1),
                         algorithmic_tflop_per_image=round(fl / 1e12, 3), tflops=round(4 * fl * K / el / 1e12, 1))
    classes = None
    if not args.no_profile and rank == 0:
        _lib.prof_begin()
        for _ in range(K):
            vae.decode(z)
        classes = _lib.prof_end()
    if rank != 0:
        return
    out = {"metric": "VAE decode images/sec, 4-item outfit @ 64x64x4 latent -> 512x512", "value": res["decode"]["images_per_s"], "unit": "images/s",
           "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": res["decode"]["ms_per_outfit"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "SURVEY 8f-1: AutoencoderKL (SD VAE shape, 83.65 M parameters), one outfit = 4 images per step per GPU"},
           "decode": res["decode"], "encode": res["encode"], "hbm_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}
    if classes is not None:
        out["decode_kernel_classes"] = {c: dict(launches_per_step=v["launches"] // K, ms_per_step=round(v["ms"] / K, 3),
                                                tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] and v["ms"] > 0 else None)
                                        for c, v in classes.items() if v["launches"]}
    if not args.no_cpu_baseline and world == 1:
        from oracle import vae_ref
        torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
        p = vae_ref.init_params(vae_ref.SD_VAE, seed=0)
        zc = torch.randn(1, 4, 64, 64)
        with torch.no_grad():
            t0 = time.perf_counter()
            vae_ref.decode(p, vae_ref.SD_VAE, zc)
            t = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=round(1.0 / t, 3), unit="images/s", cores=torch.get_num_threads(), kind="port",
                                   sample=f"oracle fp32 decode of one 64x64x4 latent = {t:.2f}s")
    print(json.dumps(out), flush=True)


def run_clip(args, da, _lib, ddist, rank, world, dev):
    """the frozen CLIP text encoder in front of the denoising path.  One step = what a run pays ONCE: the closed prompt set of
    the dataset -- 50 category prompts + the empty prompt (data_utils.py:96-111, difashion.py:226-234) = 51 sequences of 77 tokens -- through
    the HIP encoder (csrc/clip.hip, fp32 on v_mfma_f32_16x16x4_f32), i.e. PromptTable.build.  Random CLIP ViT-L/14 (SD-1.5) or OpenCLIP ViT-H/14
    (--config sd2base) shape weights, prompt-like token ids.  Roofline: the fp32 matrix instruction, 256 FLOP / clk / CU = 157.3 TFLOP/s."""
    from oracle import clip_ref          # shapes + synthetic ids only (test infrastructure); timed further down as the cpu_baseline
    cfg = clip_ref.SD2_CLIP if args.config == "sd2base" else clip_ref.SD15_CLIP
    m = da.CLIPTextModel(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                         num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads, hidden_act=cfg.hidden_act,
                         eos_token_id=cfg.eos_token_id, init_seed=0).to(dev).eval().requires_grad_(False)
    B, T = 51, 77
    ids = clip_ref.prompt_like_ids(cfg, B, T, seed=rank).to(dev)
    K, W = args.steps, args.warmup
    for _ in range(W):
        m(ids)
    torch.cuda.synchronize(); ddist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        out = m(ids)[0]
    torch.cuda.synchronize(); ddist.barrier(); torch.cuda.synchronize()
    el = ddist.max_over_ranks(time.perf_counter() - t0)
    assert torch.isfinite(out).all()
    D, I, L, H = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.num_attention_heads
    M = B * T
    gemm_flops = 2.0 * M * L * (4 * D * D + 2 * D * I)                       # q / k / v / out projections + the two MLP linears
    attn_flops = 2.0 * B * H * L * 2 * (T * (T + 1) / 2) * (D // H)          # causal scores + PV (VALU fp32)
    # HIP events on the launch stream around the K timed encodes of a second pass (the whole encoder is one stream of launches)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        m(ids)
    e1.record()
    torch.cuda.synchronize()
    ev_ms = e0.elapsed_time(e1) / K
    if rank != 0:
        return
    FP32_MFMA_PEAK = 157.3
    ach = (gemm_flops + attn_flops) / (ev_ms * 1e-3) / 1e12
    rec = {"metric": "CLIP text encoder, prompt tables/sec (51 prompts x 77 tokens)", "value": round(world * K / el, 3), "unit": "tables/s",
           "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(el * 1e3 / K, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"SURVEY 8f-2: {('OpenCLIP ViT-H/14, 23 layers x 1024' if args.config == 'sd2base' else 'CLIP ViT-L/14, 12 layers x 768')} text tower, "
                                  "the dataset's closed prompt set (50 category prompts + the empty prompt) = 51 x 77 tokens per step: what a run pays once "
                                  "(PromptTable.build)", "prompts": B, "tokens": T, "prompts_per_s": round(world * K * B / el, 1)},
           "roofline": {"bound": "mfma", "kernel": "clip_gemm_f32_kernel (v_mfma_f32_16x16x4_f32) + clip_attention_kernel", "achieved": round(ach, 2),
                        "peak": FP32_MFMA_PEAK, "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK, 4), "traffic": None,
                        "algorithmic_tflop_per_step": round((gemm_flops + attn_flops) / 1e12, 4), "event_ms_per_step": round(ev_ms, 3),
                        "note": "HIP events on the launch stream around the K encodes; fp32 matrix peak 256 FLOP/clk/CU x 256 CUs x 2.4 GHz"}}
    if not args.no_cpu_baseline and world == 1:
        torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
        p = clip_ref.init_params(cfg, 0)
        ids_c = ids.cpu()
        with torch.no_grad():
            clip_ref.clip_text_forward(p, cfg, ids_c[:4])
            t0 = time.perf_counter()
            clip_ref.clip_text_forward(p, cfg, ids_c)
            t = time.perf_counter() - t0
        rec["cpu_baseline"] = dict(value=round(1.0 / t, 3), unit="tables/s", cores=torch.get_num_threads(), kind="port",
                                   sample=f"oracle/clip_ref.py fp32 (pinned to transformers.CLIPTextModel) on the same 51 x 77 ids = {t:.2f}s")
    print(json.dumps(rec), flush=True)


def pmc_traffic(dtype="bf16"):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary (profiles/rNN/pmc_traffic.json for the bf16
    walk, profiles/rNN/pmc_traffic_fp8.json for --dtype fp8: each measured ON ITS OWN walk by scripts/profile_round.sh with separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command, FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md).  PMC
    counters cannot be read from inside the timed process.  A summary of another walk is never substituted: null instead."""
    import glob
    name = "pmc_traffic.json" if dtype == "bf16" else f"pmc_traffic_{dtype}.json"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        # the summary names the kernel sources it was measured on: a stale one (kernels edited since) is not reported
        if d.get("kernel_source_hash") != kernel_source_hash():
            return None, os.path.relpath(files[-1], ROOT) + " (STALE: kernel sources changed since it was measured; traffic withheld)"
        return float(d["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def per_launch_bound(dump_path, steps):
    """Per kernel class, from the per-launch dump of dfh_prof_end: sum of max(flops / MFMA peak, bytes / HBM peak) (the time a
    launch cannot beat on this part, nominal peaks) and of the measured durations, per step; launches counted by their bound."""
    out, table = {}, {}
    try:
        with open(dump_path) as f:
            for line in f:
                cls, flops, nbytes, ms = line.split()
                g = table.setdefault((cls, flops, nbytes), [0, 0.0]); g[0] += 1; g[1] += float(ms)
                t_mfma = float(flops) / (MFMA_BF16_PEAK * 1e12) * 1e3
                t_hbm = float(nbytes) / (HBM_PEAK * 1e9) * 1e3
                r = out.setdefault(cls, dict(attainable_ms=0.0, measured_ms=0.0, hbm_bound=0, mfma_bound=0, aux_launches=0, aux_ms=0.0))
                r["attainable_ms"] += max(t_mfma, t_hbm); r["measured_ms"] += float(ms)
                if float(flops) == 0.0:       # launches without multiply-adds inside a GEMM class: the Winograd transform kernels
                    r["aux_launches"] += 1; r["aux_ms"] += float(ms)
                r["hbm_bound" if t_hbm >= t_mfma else "mfma_bound"] += 1
        os.remove(dump_path)
    except OSError:
        return {}
    if os.environ.get("DFH_PROF_TABLE"):          # per-shape table (diagnosis): launches with equal class / flops / bytes grouped
        with open(os.environ["DFH_PROF_TABLE"], "w") as f:
            f.write(f"{'class':16s} {'GFLOP':>9s} {'MB':>8s} {'n/step':>6s} {'avg us':>8s} {'ms/step':>8s} {'TFLOP/s':>8s} {'GB/s':>7s} {'bound us':>8s}\n")
            for (cls, flops, nbytes), (n, ms) in sorted(table.items(), key=lambda kv: -kv[1][1]):
                fl, by, us = float(flops), float(nbytes), ms / n * 1e3
                lb = max(fl / (MFMA_BF16_PEAK * 1e12), by / (HBM_PEAK * 1e9)) * 1e6
                f.write(f"{cls:16s} {fl / 1e9:9.2f} {by / 1e6:8.1f} {n / steps:6.1f} {us:8.1f} {ms / steps:8.3f} {fl / us / 1e6:8.1f} {by / us / 1e3:7.0f} {lb:8.1f}\n")
    for r in out.values():
        r["attainable_ms"] /= steps; r["measured_ms"] /= steps
        r["hbm_bound"] //= steps; r["mfma_bound"] //= steps
        r["aux_launches"] //= steps; r["aux_ms"] /= steps
    return out


def kernel_source_hash():
    """sha256 over the HIP kernel sources: ties a committed PMC summary to the code it was measured on."""
    import glob, hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "difashion_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "difashion_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher environment: start N child processes of this same script, one rank per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set the way `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` sets them),
    relay rank 0's single JSON line on stdout and return non-zero when any rank fails or the job exceeds DFH_BENCH_TIMEOUT seconds
    (default 1800).  The parent itself never initialises the GPU, and a retry starts FRESH child processes (nothing here ever re-execs
    a process that has touched HIP).  Children never outlive the parent's wait loop: try / finally kills and reaps them."""
    import socket
    import subprocess
    import tempfile
    timeout = float(os.environ.get("DFH_BENCH_TIMEOUT", 1800))
    for attempt in range(3):
        with socket.socket() as s:                      # a free rendezvous port on the loopback interface (bind-then-reuse can race with
            s.bind(("127.0.0.1", 0))                    # another job on the box: an EADDRINUSE failure below is retried on a new port)
            port = s.getsockname()[1]
        procs, errs = [], []
        with tempfile.TemporaryFile("w+") as out0:
            try:
                for r in range(n):
                    # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC; with the legacy mode RCCL's
                    # cross-process buffer registration fails with `hipIpcGetMemHandle: invalid argument`.  The image exports it already;
                    # it is pinned here so that a caller with a scrubbed environment still gets working multi-process GPU runs.
                    env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
                    errs.append(tempfile.TemporaryFile("w+"))        # every rank's stderr is kept: a rank that hangs or dies is reported
                    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                                  stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=errs[-1]))
                # a rank that dies leaves the others waiting in a collective: the first non-zero exit ends the whole job
                bad, t0 = [], time.time()
                while not bad and any(p.poll() is None for p in procs):
                    time.sleep(0.2)
                    bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
                    if not bad and time.time() - t0 > timeout:
                        bad = [(r, "timeout") for r, p in enumerate(procs) if p.poll() is None]
            finally:
                for p in procs:                          # also on KeyboardInterrupt / any exception in the loop above
                    if p.poll() is None:
                        p.kill()
                for p in procs:
                    p.wait()
            tails = []
            for r, f in enumerate(errs):
                f.seek(0)
                tails.append(f.read())
                f.close()
            if tails and tails[0]:
                sys.stderr.write(tails[0])               # rank 0's log lines, as when it runs alone
            if bad:
                log(f"[bench] ranks failed (rank, exit code): {bad}")
                for r, _ in bad:
                    log(f"[bench] ---- stderr tail of rank {r}:\n" + tails[r][-2000:])
                if attempt < 2 and any("EADDRINUSE" in tails[r] or "address already in use" in tails[r].lower() for r, _ in bad):
                    log("[bench] rendezvous port was taken: retrying on a new port with fresh ranks")
                    continue
                return 1
            out0.seek(0)
            text = out0.read()
        lines = [l for l in text.splitlines() if l.startswith("{")]
        if len(lines) != 1:
            log(f"[bench] rank 0 printed {len(lines)} JSON lines, expected exactly one")
            return 1
        print(lines[0], flush=True)
        return 0
    return 1


def sampling_roofline(_lib, classes, bound, K, dtype, pmc_walk=True):
    """roofline object of a sampling run from the live HIP-event profile (dominant family: every conv3x3 / 1x1 / linear GEMM)."""
    gemm = {k: sum(classes[c][k] for c in ("gemm_conv3x3", "gemm_linear")) for k in ("launches", "ms", "flops", "bytes")}
    achieved = gemm["flops"] / (gemm["ms"] * 1e-3) / 1e12
    aux_n = sum(bound.get(c, {}).get("aux_launches", 0) for c in ("gemm_conv3x3", "gemm_linear"))
    aux_ms = sum(bound.get(c, {}).get("aux_ms", 0.0) for c in ("gemm_conv3x3", "gemm_linear"))
    # flops = the REFERENCE algorithm's; the upsampler convs (phase planes) and the Winograd convs execute fewer
    saved = float(_lib.raw().dfh_prof_saved_flops())
    executed = (gemm["flops"] - saved) / (gemm["ms"] * 1e-3) / 1e12
    # the committed PMC summaries were measured on the configs[1] walk (SD-1.5 shape, one outfit): another shape / batch reports null, never that figure
    traffic, traffic_src = pmc_traffic(dtype) if pmc_walk else (None, "no PMC summary for this workload (profiles/rNN/pmc_traffic*.json are the SD-1.5, one-outfit walk)")
    roofline = dict(bound="mfma", kernel="gemm_bf16_kernel + gemm_wide_kernel + mlp2_fused_kernel (conv3x3 + 1x1 + linear: one implicit GEMM, all tile variants; "
                                         "the Winograd transform launches are timed with the convs they belong to)",
                    achieved=round(achieved, 1), peak=MFMA_BF16_PEAK, unit="TFLOP/s", frac=round(achieved / MFMA_BF16_PEAK, 4),
                    traffic=traffic, traffic_unit="HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE), measured on this walk's own dtype",
                    traffic_source=traffic_src,
                    algorithmic_bytes_per_launch=round(gemm["bytes"] / max(1, gemm["launches"])),
                    sustained_mfma_only_peak=MFMA_BF16_SUSTAINED, frac_of_sustained=round(achieved / MFMA_BF16_SUSTAINED, 4),
                    # the GEMM kernel's own launches (what rocprofv3 lists under gemm_bf16_kernel / gemm_wide_kernel): the conv class also
                    # holds the Winograd transform launches (no multiply-adds), whose time stays inside `achieved`
                    launches_per_step=gemm["launches"] // K - aux_n,
                    avg_launch_us=round((gemm["ms"] * 1e3 / K - aux_ms * 1e3) / max(1, gemm["launches"] // K - aux_n), 2),
                    transform_launches_per_step=aux_n, transform_ms_per_step=round(aux_ms, 3),
                    algorithmic_tflop_per_step=round(gemm["flops"] / K / 1e12, 3),
                    executed_tflop_per_step=round((gemm["flops"] - saved) / K / 1e12, 3),
                    executed_tflops=round(executed, 1), executed_frac=round(executed / MFMA_BF16_PEAK, 4),
                    note="achieved / frac count the reference algorithm's multiply-adds (direct 3x3 convs); executed_* what the MFMA pipe "
                         "ran after the phase-plane upsamplers (4/9) and the Winograd F(2x2,3x3) convs of the 16x16 / 8x8 levels (16/36)")
    # secondary kernels, same live HIP-event timing: attention against the bf16 MFMA peak, GroupNorm against HBM, and (fp8
    # runs) the e4m3 GEMM class against the fp8 MFMA peak
    sec = {}
    def tf(c):
        return classes[c]["flops"] / (classes[c]["ms"] * 1e-3) / 1e12
    if classes["attention"]["ms"] > 0:
        sec["attention"] = dict(bound="mfma", kernel="attention_x32_kernel + attention_kernel", achieved=round(tf("attention"), 1),
                                peak=MFMA_BF16_PEAK, unit="TFLOP/s", frac=round(tf("attention") / MFMA_BF16_PEAK, 4))
    if classes["groupnorm"]["ms"] > 0:
        gbs = classes["groupnorm"]["bytes"] / (classes["groupnorm"]["ms"] * 1e-3) / 1e9
        sec["groupnorm"] = dict(bound="hbm", kernel="gn_stats_kernel + gn_apply_kernel + gn_small_kernel", achieved=round(gbs, 1),
                                peak=HBM_PEAK, unit="GB/s", frac=round(gbs / HBM_PEAK, 4))
    if classes.get("gemm_linear_fp8", {}).get("ms", 0) > 0:
        sec["gemm_linear_fp8"] = dict(bound="mfma", kernel="gemm_fp8_kernel (v_mfma_scale_f32_32x32x64_f8f6f4, e4m3)",
                                      achieved=round(tf("gemm_linear_fp8"), 1), peak=MFMA_FP8_PEAK, unit="TFLOP/s",
                                      frac=round(tf("gemm_linear_fp8") / MFMA_FP8_PEAK, 4),
                                      launches_per_step=classes["gemm_linear_fp8"]["launches"] // K,
                                      ms_per_step=round(classes["gemm_linear_fp8"]["ms"] / K, 3))
    roofline["secondary"] = sec
    # the family is a MIX of MFMA-bound and HBM-bound launches (a 65536 x 320 x 320 linear moves 126 MB for 13 GFLOP: 16 us at
    # 8 TB/s, 5 us at 2.5 PFLOP/s), so beside the family-wide flop rate: sum over launches of max(flops / MFMA peak, algorithmic
    # bytes / HBM peak) against the summed measured durations
    gb = [bound.get(c) for c in ("gemm_conv3x3", "gemm_linear") if bound.get(c)]
    if gb:
        att = sum(b["attainable_ms"] for b in gb); meas = sum(b["measured_ms"] for b in gb)
        roofline["per_launch_bound"] = dict(
            attainable_ms_per_step=round(att, 3), measured_ms_per_step=round(meas, 3), frac=round(att / meas, 4),
            hbm_bound_launches_per_step=sum(b["hbm_bound"] for b in gb), mfma_bound_launches_per_step=sum(b["mfma_bound"] for b in gb),
            note="sum over launches of max(flops / 2500 TFLOP/s, algorithmic bytes / 8000 GB/s) over the sum of measured durations")
    return roofline


def measure_sampling(da, _lib, ddist, unet, enc, dev, rank, K, W, outfits=1, dtype="bf16", profile=True):
    """W warm-up + K timed steps of the sampling loop for ``outfits`` outfits on this GPU (barrier + device sync on both sides, MAX over
    the ranks), then -- on rank 0 -- a second, profiled pass of the same K steps with HIP events on every launch."""
    cross = unet.config.cross_attention_dim
    sampler = da.OutfitSampler(unet, enc, da.DDIMScheduler())
    inp = outfit_inputs(dev, cross, rank, outfits)

    def run_steps(n, offset):
        for i in range(n):
            sampler.step((offset + i) % 50)

    sampler.prepare(num_inference_steps=50, cate_scale=12.0, hist_scale=4.0, mutual_scale=5.0, eta=0.1, **inp)
    run_steps(W, 0)
    torch.cuda.synchronize()
    ddist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(K, W)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0                     # this rank's own K steps, before it waits for the others
    ddist.barrier()
    torch.cuda.synchronize()
    elapsed = ddist.max_over_ranks(time.perf_counter() - t0)
    own_max, own_min = ddist.max_over_ranks(own), min_over_ranks(ddist, own)
    assert torch.isfinite(sampler.latents).all(), "non-finite latents"
    res = dict(elapsed=elapsed, own_min=own_min, own_max=own_max, roofline=None, classes=None, bound={}, prof_ms=None)
    if profile and rank == 0:
        torch.cuda.synchronize()
        _lib.prof_begin()
        t1 = time.perf_counter()
        run_steps(K, W)
        import tempfile
        dump = os.path.join(tempfile.gettempdir(), f"dfh_prof_dump_{os.getpid()}.txt")
        os.environ["DFH_PROF_DUMP"] = dump                       # one line per launch: class, flops, bytes, ms
        classes = _lib.prof_end()
        os.environ.pop("DFH_PROF_DUMP", None)
        res["prof_ms"] = (time.perf_counter() - t1) * 1e3 / K
        res["bound"] = per_launch_bound(dump, K)
        res["classes"] = classes
        res["roofline"] = sampling_roofline(_lib, classes, res["bound"], K, dtype,
                                            pmc_walk=(outfits == 1 and cross == 768))      # the PMC summaries are the SD-1.5, one-outfit walk
    unet.end_run()
    return res


def class_table(classes, bound, K):
    def rate(c, key, scale):
        return round(classes[c][key] / (classes[c]["ms"] * 1e-3) / scale, 1) if classes[c]["ms"] > 0 else None
    return {c: dict(launches_per_step=v["launches"] // K, ms_per_step=round(v["ms"] / K, 3),
                    tflops=rate(c, "flops", 1e12) if v["flops"] else None, algorithmic_GBps=rate(c, "bytes", 1e9),
                    roofline_ms_per_step=round(bound[c]["attainable_ms"], 3) if c in bound else None)
            for c, v in classes.items() if v["launches"]}


def secondary_configs(args, da, _lib, ddist, unet, enc, dev):
    """Short legs of the other BASELINE configurations, run on rank 0 of a one-GPU default invocation AFTER the headline's timed
    region and profile pass (they cannot touch it).  Each leg is fenced: a failure is recorded under its key, the headline still prints."""
    out = {}
    t_all = time.time()
    # ---- configs[4]: the same sampler with the transformer linears in e4m3 (one GPU; the 8-GPU half of configs[4] is outfit replicas)
    try:
        unet.enable_fp8(True)
        unet.pack(force=True)
        r = measure_sampling(da, _lib, ddist, unet, enc, dev, 0, 10, 3, outfits=1, dtype="fp8")
        rf = r["roofline"]
        out["configs[4]"] = dict(workload="BASELINE configs[4] on one GPU: the configs[1] sampler with enable_fp8() (all nine linears / 1x1 convs of every "
                                          "transformer block in e4m3, U-Net batch 16)", steps=10, warmup=3,
                                 ms_per_step=round(r["elapsed"] * 1e3 / 10, 3), steps_per_s=round(10 / r["elapsed"], 3),
                                 dtype="fp8 e4m3 transformer linears + bf16 convs / attention products",
                                 roofline=dict(bound=rf["bound"], achieved=rf["achieved"], peak=rf["peak"], unit=rf["unit"], frac=rf["frac"],
                                               traffic=rf["traffic"], traffic_source=rf["traffic_source"],
                                               secondary=dict(gemm_linear_fp8=rf["secondary"].get("gemm_linear_fp8"),
                                                              attention=rf["secondary"].get("attention"))),
                                 kernel_classes=class_table(r["classes"], r["bound"], 10))
    except Exception as e:
        out["configs[4]"] = {"error": repr(e)}
    finally:
        unet.enable_fp8(False)
        unet.pack(force=True)
    # ---- the reference's own inference batch: 4 outfits per call (GOR, inf4eval.py:521-524) -> U-Net batch 64
    try:
        r = measure_sampling(da, _lib, ddist, unet, enc, dev, 0, 5, 2, outfits=4, dtype="bf16")
        rf = r["roofline"]
        out["inference_batch_64"] = dict(workload="the reference's own inference call (inf4eval.py:521-524, GOR): 4 outfits x 4 items x 4 guidance branches "
                                                  "-> U-Net batch 64, bf16, same sampler", steps=5, warmup=2, unet_batch=64,
                                         ms_per_step=round(r["elapsed"] * 1e3 / 5, 3), outfit_steps_per_s=round(4 * 5 / r["elapsed"], 3),
                                         ms_per_step_per_outfit=round(r["elapsed"] * 1e3 / 5 / 4, 3),
                                         roofline=dict(bound=rf["bound"], achieved=rf["achieved"], peak=rf["peak"], unit=rf["unit"], frac=rf["frac"],
                                                       executed_frac=rf["executed_frac"], per_launch_bound=rf.get("per_launch_bound"),
                                                       secondary=rf["secondary"]),
                                         kernel_classes=class_table(r["classes"], r["bound"], 5))
    except Exception as e:
        out["inference_batch_64"] = {"error": repr(e)}
    return out, time.time() - t_all


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="sd15", choices=["sd15", "sd2base"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short configs[4] / inference-batch-64 / configs[2] legs the default one-GPU invocation runs after the headline")
    ap.add_argument("--mode", default="sample", choices=["sample", "train", "vae", "clip"],
                    help="sample: the headline metric (default); train: BASELINE configs[2]/[3] training step; vae: SURVEY 8f-1; clip: SURVEY 8f-2 "
                         "(the prompt table through the HIP CLIP text encoder)")
    ap.add_argument("--outfits", type=int, default=8, help="--mode train: outfits per GPU per step")
    ap.add_argument("--outfits-per-gpu", type=int, default=1,
                    help="--mode sample: outfits denoised together on each GPU (1 = BASELINE configs[1], U-Net batch 16; 4 = the reference's own "
                         "inference call, inf4eval.py:521-524, U-Net batch 64 -- a SECONDARY line, never the headline)")
    ap.add_argument("--wire", default=None, choices=["fp32", "bf16"],
                    help="--mode train: gradient exchange format.  fp32 (default): one RCCL all-reduce per range, the reference's DDP semantics; "
                         "bf16 (opt-in): all_to_all + fp32 accumulate + all_gather, half the bytes per xGMI link, gradients rounded to bf16 on the wire")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: BASELINE configs[4] -- every linear / 1x1 convolution of the transformer blocks (proj_in, q|k, v, to_out, "
                         "cross q, cross to_out, the GEGLU pair, proj_out) in e4m3 on the block-scaled MFMA; 3x3 convs stay bf16")
    ap.add_argument("--fp8-attention", action="store_true",
                    help="with --dtype fp8: also the self-attention products QK^T / PV on the e4m3 MFMA (opt-in: measured slower, profiles/r04)")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU-only check of the --gpus N launcher flow (tests/test_host_logic_cpu.py, world 8): every rank joins a gloo group "
                         "from the RANK / WORLD_SIZE / MASTER_* environment launch_ranks gives it, one all-reduce counts the ranks, rank 0 prints "
                         "one JSON line.  Touches no GPU and measures nothing")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  Nothing here has touched the GPU yet (no HIP call, no
        # torch.cuda.is_available()), and it never will: it starts N fresh ranks and relays rank 0's line.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    if args.selftest_launcher:
        from difashion_amd import dist as ddist
        rank, world, local = ddist.init("gloo")
        seen = int(round(ddist.sum_over_ranks(1.0)))
        slowest = ddist.max_over_ranks(float(rank))               # the max-over-ranks reduction the timed region closes with
        ddist.barrier()
        if rank == 0:
            print(json.dumps({"selftest": "launcher", "n_gpus": args.gpus, "world_size": world, "ranks_seen": seen, "max_rank": int(slowest),
                              "local_rank": local, "collective_backend": "gloo", "rccl_ranks": None}), flush=True)
        return

    import difashion_amd as da
    from difashion_amd import _lib, dist as ddist

    # RCCL ("nccl") under the launcher; DFH_DIST_BACKEND=gloo lets the multi-rank flow be exercised on a box with fewer GPUs
    # than ranks (tests/test_gpu_ddp.py: two ranks sharing the one device) -- never the measured configuration
    backend = os.environ.get("DFH_DIST_BACKEND", "nccl")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    local_dev = int(os.environ.get("LOCAL_RANK", 0)) % max(1, torch.cuda.device_count()) if backend != "nccl" else None
    # DFH_DIST_SINGLE_RANK=1: a world of ONE rank that still runs the gradient exchange over RCCL (what the data-parallel machinery
    # costs on one GPU when the wire is free: --mode train prints comm_exposed_ms for it)
    single_rank = os.environ.get("DFH_DIST_SINGLE_RANK") == "1"
    rank, world, local = ddist.init(backend if (args.gpus > 1 or single_rank) else None)
    if world != args.gpus:
        log(f"[bench] WARNING: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    if local_dev is not None:
        local = local_dev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # one all-reduce of ones over the job's process group before anything is timed: the number of ranks the collective backend really
    # connected (RCCL over xGMI under "nccl"); it goes into the line as rccl_ranks
    args.ranks_seen = int(round(ddist.sum_over_ranks(1.0))) if world > 1 else 1
    args.backend = backend if world > 1 else None
    if args.ranks_seen != world:
        raise SystemExit(f"bench.py: the all-reduce saw {args.ranks_seen} ranks, WORLD_SIZE is {world}")

    if args.mode == "train":
        return run_train(args, da, _lib, ddist, rank, world, dev)
    if args.mode == "vae":
        return run_vae(args, da, _lib, ddist, rank, world, dev)
    if args.mode == "clip":
        return run_clip(args, da, _lib, ddist, rank, world, dev)
    unet, enc = build_models(dev, args.config)
    if args.dtype == "fp8":
        unet.enable_fp8(True, attention=args.fp8_attention)
        unet.pack(force=True)
    K, W = args.steps, args.warmup
    nout = max(1, args.outfits_per_gpu)
    r = measure_sampling(da, _lib, ddist, unet, enc, dev, rank, K, W, outfits=nout, dtype=args.dtype, profile=not args.no_profile)
    elapsed, roofline, classes, bound = r["elapsed"], r["roofline"], r["classes"], r["bound"]

    if world > 1:
        ddist.barrier()
    if rank != 0:
        return
    value = world * nout * K / elapsed
    if nout == 1:
        workload = ("BASELINE configs[1]" if args.dtype == "bf16" else "BASELINE configs[4] on one GPU (fp8 transformer linears"
                    + (" + fp8 attention products)" if args.fp8_attention else ")")) + \
                   ": one 4-item outfit, CFG on (4 branches) -> U-Net batch 16, DDIM-50 schedule, " \
                   f"{args.config} shape in_channels=8, 64x64x4 latents, 77 text tokens; one outfit per GPU"
    else:
        workload = (f"SECONDARY line (not BASELINE configs[1]): {nout} 4-item outfits per GPU per call (the reference's inference call, "
                    f"inf4eval.py:521-524), CFG on -> U-Net batch {16 * nout}, DDIM-50 schedule, {args.config} shape, 64x64x4 latents")
    out = {
        "metric": "U-Net denoise steps/sec, 4-item outfit @ 64x64x4 latent", "value": round(value, 3), "unit": "steps/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed * 1e3 / K, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if args.dtype == "bf16" else
                 ("fp8 e4m3 (all nine linears / 1x1 convs of every transformer block, GEGLU hidden tensor in e4m3 + E8M0 block scales"
                  + (", self-attention QK^T / PV" if args.fp8_attention else "") + ") + bf16 (3x3 convs"
                  + ("" if args.fp8_attention else ", attention products") + ", residual stream)"),
        "data": "synthetic",
        "config": {"workload": workload, "unet_batch": 16 * nout, "latent": "64x64x4",
                   "parallelism": f"outfit-replicas x{world} (no data-path collective)",
                   # the two guidance branches that differ only in their prompt (difashion.py:388-427) share conv_in, the first resnet and the
                   # first block up to its self-attention (bf16 walk; DFH_CFG_DEDUP=0 turns it off): images whose prefix is computed once
                   "cfg_shared_prefix_images": (4 * nout if (args.dtype == "bf16" and os.environ.get("DFH_CFG_DEDUP", "1") != "0") else 0)},
        "roofline": roofline,
        # ranks counted by an all-reduce of ones on the job's process group (RCCL when the backend is "nccl"; null under the gloo test backend)
        "rccl_ranks": args.ranks_seen if args.backend in (None, "nccl") else None, "collective_backend": args.backend,
        "rccl_ranks_match_n_gpus": (args.ranks_seen == world) if args.backend in (None, "nccl") else None,
    }
    if nout > 1:
        out["ms_per_step_per_outfit"] = round(elapsed * 1e3 / K / nout, 3)
    if world > 1:       # a slow rank must be visible from this one record
        out["rank_ms_per_step"] = dict(min=round(r["own_min"] * 1e3 / K, 3), max=round(r["own_max"] * 1e3 / K, 3),
                                       note="each rank's own K steps up to its device sync, before the closing barrier")
    if classes is not None:
        out["kernel_classes"] = class_table(classes, bound, K)
        out["profiled_pass_ms_per_step"] = round(r["prof_ms"], 3)
        attn = classes["attention"]
        if attn["ms"] > 0:
            out["attention_mfma_frac"] = round(attn["flops"] / (attn["ms"] * 1e-3) / 1e12 / MFMA_BF16_PEAK, 4)
            out["attention_frac_of_sustained"] = round(attn["flops"] / (attn["ms"] * 1e-3) / 1e12 / MFMA_BF16_SUSTAINED, 4)
        gn = classes["groupnorm"]
        if gn["ms"] > 0:
            out["groupnorm_hbm_frac"] = round(gn["bytes"] / (gn["ms"] * 1e-3) / 1e9 / HBM_PEAK, 4)
    # ---- driver-witnessed legs of the other configurations (default one-GPU bf16 invocation only; after everything the headline measures)
    if world == 1 and not args.no_secondary and not args.no_profile and args.dtype == "bf16" and nout == 1 and args.config == "sd15":
        sec, took = secondary_configs(args, da, _lib, ddist, unet, enc, dev)
        del unet, enc
        import gc
        gc.collect(); torch.cuda.empty_cache()
        t1 = time.time()
        try:
            args.wire = None
            tr = measure_train(args, da, _lib, ddist, 0, 1, dev, 4, 2, profile=True)
            sec["configs[2]"] = dict(workload=tr["config"]["workload"], steps=4, warmup=2, ms_per_step=tr["ms_per_step"], items_per_s=tr["value"],
                                     unet_batch=tr["config"]["unet_batch"], loss=tr["loss"], hbm_GiB=tr["hbm_GiB"],
                                     mfma_frac_whole_step=tr.get("mfma_frac_whole_step"), roofline=tr["roofline"])
        except Exception as e:
            sec["configs[2]"] = {"error": repr(e)}
        # ---- SD-2-base, the reference's OWN default model (train.py:44, inf4eval.py:65: linear projections, 1024-wide text states, head dim 64
        #      at every level): the configs[1] sampler on that shape, bf16, so that the default model is driver-witnessed too
        try:
            gc.collect(); torch.cuda.empty_cache()
            u2, e2 = build_models(dev, "sd2base")
            r2 = measure_sampling(da, _lib, ddist, u2, e2, dev, 0, 10, 3, outfits=1, dtype="bf16", profile=False)
            sec["sd2base_bf16"] = dict(workload="the configs[1] sampler on the SD-2-base shape (the reference's default pretrained_model_name_or_path, "
                                                "train.py:44): one 4-item outfit, CFG on -> U-Net batch 16, DDIM-50 schedule, bf16",
                                       steps=10, warmup=3, ms_per_step=round(r2["elapsed"] * 1e3 / 10, 3), steps_per_s=round(10 / r2["elapsed"], 3))
            del u2, e2
        except Exception as e:
            sec["sd2base_bf16"] = {"error": repr(e)}
        sec["seconds_spent"] = round(took + time.time() - t1, 1)
        out["secondary_configs"] = sec
    if not args.no_cpu_baseline and world == 1:
        try:
            # host threads: the cores this process may run on, capped at 16 (the fp32 oracle's convs do not
            # scale past that; 256 oversubscribed threads measured 30x slower)
            out["cpu_baseline"] = cpu_baseline(min(16, len(os.sched_getaffinity(0))))
        except Exception as e:  # the GPU number must still be reported
            out["cpu_baseline"] = {"error": repr(e)}
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
