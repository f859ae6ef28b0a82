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
                   this box's host cores on a bounded sample.
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


def outfit_inputs(dev, cross_dim, rank):
    """Synthetic iFashion-shaped inputs: one 4-item outfit, every slot generated (GOR)."""
    def rn(seed, *shape):
        return torch.randn(*shape, generator=torch.Generator().manual_seed(seed + 1000 * rank)).to(dev)
    return dict(olists=torch.zeros(1, 4, dtype=torch.long), all_latents=rn(122, 4, 4, 64, 64) * 0.18215,
                init_latents=rn(123, 4, 4, 64, 64), hist_latents=rn(124, 4, 4, 64, 64) * 0.18215,
                null_latent=rn(125, 4, 64, 64) * 0.18215, category_prompts=rn(126, 4, 77, cross_dim),
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


def run_train(args, da, _lib, ddist, rank, world, dev):
    """BASELINE configs[2]/[3]: one optimisation step = loss forward + native backward + RCCL gradient all-reduce + clip/AdamW
    + EMA over a per-GPU batch of 8 outfits x 4 items (weak scaling: global batch = 32 x N items)."""
    unet, enc = build_models(dev, args.config)
    unet.train(); enc.train()
    opt = da.FusedAdamW(list(unet.parameters()) + list(enc.parameters()), lr=1e-5, weight_decay=1e-2, max_grad_norm=1.0)
    ddist.broadcast_parameters(opt.flat_param)          # replicas start identical (DDP's construction-time broadcast)
    ema = da.EMAModel(unet.parameters())
    sched = da.DDIMScheduler()
    kw = train_inputs(dev, unet.config.cross_attention_dim, rank, args.outfits)
    K, W = args.steps, args.warmup
    if args.wire is None:
        args.wire = "bf16" if ddist.active() else "fp32"
    unet.grad_wire_dtype = args.wire
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
    torch.cuda.synchronize(); ddist.barrier(); torch.cuda.synchronize()
    elapsed = ddist.max_over_ranks(time.perf_counter() - t0)
    comm_exposed_ms = None
    if exposed and all(e is not None for e in exposed):
        comm_exposed_ms = ddist.max_over_ranks(sum(e0.elapsed_time(e1) for e0, e1 in exposed) / len(exposed))
    assert torch.isfinite(loss), "non-finite loss"
    classes = None
    if not args.no_profile and rank == 0 and world == 1:
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
    if rank != 0:
        return
    items = args.outfits * 4
    out = {"metric": "training items/sec, fwd+bwd+AdamW, 8 outfits x 4 items per GPU @ 64x64x4 latent", "value": round(world * items * K / elapsed, 2),
           "unit": "items/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed * 1e3 / K, 2),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"BASELINE configs[2]: training step, {args.outfits} outfits x 4 items per GPU, {args.config} shape in_channels=8, "
                                  "min-SNR MSE loss, mutual + history conditioning, clip 1.0 + AdamW + EMA",
                      "unet_batch": items, "parallelism": f"data-parallel x{world}, gradients all-reduced over RCCL in 256 MB ranges of the packed arena, overlapped with the backward walk"},
           "loss": round(float(loss), 5), "hbm_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 1),
           # time per step the compute stream waited for the side-stream gradient exchange = the part of the exchange the backward
           # walk did not hide (max over ranks; null on one GPU: no exchange)
           "comm_exposed_ms": None if comm_exposed_ms is None else round(comm_exposed_ms, 3), "grad_wire": args.wire,
           "rccl_ranks": args.ranks_seen if args.backend in (None, "nccl") else None, "collective_backend": args.backend}
    if world == 1 and ddist.active():      # DFH_DIST_SINGLE_RANK=1: the gradient exchange ran over RCCL in a world of one rank
        out["single_rank_collectives"] = True
        out["collective_backend"] = backend_name()
    if classes is not None:
        tot_f = sum(v["flops"] for v in classes.values())
        out["kernel_classes"] = {c: dict(launches_per_step=v["launches"] // K, ms_per_step=round(v["ms"] / K, 3),
                                         tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] and v["ms"] > 0 else None,
                                         algorithmic_GBps=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None)
                                 for c, v in classes.items() if v["launches"]}
        out["algorithmic_tflop_per_step"] = round(tot_f / K / 1e12, 2)
        out["mfma_frac_whole_step"] = round(tot_f / K / (elapsed / K) / 1e12 / MFMA_BF16_PEAK, 4)
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


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary (profiles/rNN/pmc_traffic.json,
    written by scripts/profile_round.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command,
    FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md).  PMC counters cannot be read from inside the timed process."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")))
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
    relay rank 0's single JSON line on stdout and return non-zero when any rank fails.  The parent itself never initialises the GPU
    (children are fresh processes: no exec of a process that has touched HIP)."""
    import socket
    import subprocess
    with socket.socket() as s:                      # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile
    procs = []
    with tempfile.TemporaryFile("w+") as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        # a rank that dies leaves the others waiting in a collective: the first non-zero exit ends the whole job
        bad = []
        while not bad and any(p.poll() is None for p in procs):
            time.sleep(0.2)
            bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad:
            log(f"[bench] ranks failed (rank, exit code): {bad}")
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
            return 1
        out0.seek(0)
        text = out0.read()
    lines = [l for l in text.splitlines() if l.startswith("{")]
    if len(lines) != 1:
        log(f"[bench] rank 0 printed {len(lines)} JSON lines, expected exactly one")
        return 1
    print(lines[0], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="sd15", choices=["sd15", "sd2base"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--mode", default="sample", choices=["sample", "train", "vae"],
                    help="sample: the headline metric (default); train: BASELINE configs[2]/[3] training step; vae: SURVEY 8f-1")
    ap.add_argument("--outfits", type=int, default=8, help="--mode train: outfits per GPU per step")
    ap.add_argument("--wire", default=None, choices=["fp32", "bf16"],
                    help="--mode train: gradient exchange format (bf16: all_to_all + fp32 accumulate + all_gather, half the bytes per link); "
                         "default: bf16 when more than one GPU takes part, fp32 (nothing is exchanged) on one")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: BASELINE configs[4] -- every linear / 1x1 convolution of the transformer blocks (proj_in, q|k, v, to_out, "
                         "cross q, cross to_out, the GEGLU pair, proj_out) in e4m3 on the block-scaled MFMA; 3x3 convs stay bf16")
    ap.add_argument("--fp8-attention", action="store_true",
                    help="with --dtype fp8: also the self-attention products QK^T / PV on the e4m3 MFMA (opt-in: measured slower, profiles/r04)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  Nothing here has touched the GPU yet (no HIP call, no
        # torch.cuda.is_available()), and it never will: it starts N fresh ranks and relays rank 0's line.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

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
    unet, enc = build_models(dev, args.config)
    if args.dtype == "fp8":
        unet.enable_fp8(True, attention=args.fp8_attention)
        unet.pack(force=True)
    cross = unet.config.cross_attention_dim
    K, W = args.steps, args.warmup
    sampler = da.OutfitSampler(unet, enc, da.DDIMScheduler())
    inp = outfit_inputs(dev, cross, rank)

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
    ddist.barrier()
    torch.cuda.synchronize()
    elapsed = ddist.max_over_ranks(time.perf_counter() - t0)
    assert torch.isfinite(sampler.latents).all(), "non-finite latents"

    roofline, classes, prof_ms = None, None, None
    if not args.no_profile and rank == 0:
        torch.cuda.synchronize()
        _lib.prof_begin()
        t1 = time.perf_counter()
        run_steps(K, W)
        import tempfile
        dump = os.path.join(tempfile.gettempdir(), f"dfh_prof_dump_{os.getpid()}.txt")
        os.environ["DFH_PROF_DUMP"] = dump                       # one line per launch: class, flops, bytes, ms
        classes = _lib.prof_end()
        os.environ.pop("DFH_PROF_DUMP", None)
        prof_ms = (time.perf_counter() - t1) * 1e3 / K
        bound = per_launch_bound(dump, K)
        gemm = {k: sum(classes[c][k] for c in ("gemm_conv3x3", "gemm_linear")) for k in ("launches", "ms", "flops", "bytes")}
        achieved = gemm["flops"] / (gemm["ms"] * 1e-3) / 1e12
        aux_n = sum(bound.get(c, {}).get("aux_launches", 0) for c in ("gemm_conv3x3", "gemm_linear"))
        aux_ms = sum(bound.get(c, {}).get("aux_ms", 0.0) for c in ("gemm_conv3x3", "gemm_linear"))
        # flops = the REFERENCE algorithm's; the upsampler convs (phase planes) and the Winograd convs execute fewer
        saved = float(_lib.raw().dfh_prof_saved_flops())
        executed = (gemm["flops"] - saved) / (gemm["ms"] * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic()
        roofline = dict(bound="mfma", kernel="gemm_bf16_kernel + gemm_wide_kernel (conv3x3 + 1x1 + linear: one implicit GEMM, all tile variants; "
                                             "the Winograd transform launches are timed with the convs they belong to)",
                        achieved=round(achieved, 1), peak=MFMA_BF16_PEAK, unit="TFLOP/s", frac=round(achieved / MFMA_BF16_PEAK, 4),
                        traffic=traffic, traffic_unit="HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", traffic_source=traffic_src,
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
                                          launches_per_step=classes["gemm_linear_fp8"]["launches"] // K)
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

    if world > 1:
        ddist.barrier()
    if rank != 0:
        return
    value = world * K / elapsed
    out = {
        "metric": "U-Net denoise steps/sec, 4-item outfit @ 64x64x4 latent", "value": round(value, 3), "unit": "steps/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed * 1e3 / K, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if args.dtype == "bf16" else
                 ("fp8 e4m3 (all nine linears / 1x1 convs of every transformer block, GEGLU hidden tensor in e4m3 + E8M0 block scales"
                  + (", self-attention QK^T / PV" if args.fp8_attention else "") + ") + bf16 (3x3 convs"
                  + ("" if args.fp8_attention else ", attention products") + ", residual stream)"),
        "data": "synthetic",
        "config": {"workload": ("BASELINE configs[1]" if args.dtype == "bf16" else "BASELINE configs[4] on one GPU (fp8 transformer linears"
                                + (" + fp8 attention products)" if args.fp8_attention else ")")) +
                               ": one 4-item outfit, CFG on (4 branches) -> U-Net batch 16, DDIM-50 schedule, "
                               f"{args.config} shape in_channels=8, 64x64x4 latents, 77 text tokens; one outfit per GPU",
                   "unet_batch": 16, "latent": "64x64x4", "parallelism": f"outfit-replicas x{world} (no data-path collective)"},
        "roofline": roofline,
        # ranks counted by an all-reduce of ones on the job's process group (RCCL when the backend is "nccl"; null under the gloo test backend)
        "rccl_ranks": args.ranks_seen if args.backend in (None, "nccl") else None, "collective_backend": args.backend,
    }
    if classes is not None:
        def rate(c, key, scale):
            return round(classes[c][key] / (classes[c]["ms"] * 1e-3) / scale, 1) if classes[c]["ms"] > 0 else None
        out["kernel_classes"] = {
            c: dict(launches_per_step=v["launches"] // K, ms_per_step=round(v["ms"] / K, 3),
                    tflops=rate(c, "flops", 1e12) if v["flops"] else None, algorithmic_GBps=rate(c, "bytes", 1e9),
                    roofline_ms_per_step=round(bound[c]["attainable_ms"], 3) if c in bound else None)
            for c, v in classes.items() if v["launches"]}
        out["profiled_pass_ms_per_step"] = round(prof_ms, 3)
        attn = classes["attention"]
        if attn["ms"] > 0:
            out["attention_mfma_frac"] = round(attn["flops"] / (attn["ms"] * 1e-3) / 1e12 / MFMA_BF16_PEAK, 4)
            out["attention_frac_of_sustained"] = round(attn["flops"] / (attn["ms"] * 1e-3) / 1e12 / MFMA_BF16_SUSTAINED, 4)
        gn = classes["groupnorm"]
        if gn["ms"] > 0:
            out["groupnorm_hbm_frac"] = round(gn["bytes"] / (gn["ms"] * 1e-3) / 1e9 / HBM_PEAK, 4)
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
