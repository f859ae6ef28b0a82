#!/usr/bin/env python3
"""Regenerates the numbers of a round's section in profiles/README.md FROM the committed artefacts of that round (bench JSON lines,
rocprofv3 kernel_stats.csv, pmc_traffic.json), so that the prose cannot drift away from the files it cites (hygiene).
Usage: python scripts/profiles_readme.py r03 [--check]     (--check: exit 1 if the committed block differs; tests/test_host_logic_cpu.py)"""
import csv, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json(path):
    try:
        lines = [l for l in open(path).read().strip().splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except OSError:
        return None


def gemm_family(stats_csv):
    """(launches, total ms, avg us) of the implicit-GEMM family in a rocprofv3 kernel_stats.csv."""
    n, ns = 0, 0.0
    for r in csv.DictReader(open(stats_csv)):
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        if name.startswith(("gemm_bf16_kernel", "gemm_wide_kernel", "mlp2_fused_kernel")):      # the fused feed-forward is part of the family (round 5)
            n += int(r["Calls"]); ns += float(r["TotalDurationNs"])
    return n, ns / 1e6, (ns / n / 1e3 if n else 0.0)


def block(tag):
    d = os.path.join(ROOT, "profiles", tag)
    out = []
    b = last_json(os.path.join(d, "bench_n1.json"))
    if b:
        rf, kc = b["roofline"], b.get("kernel_classes", {})
        out.append(f"* sampling (`{tag}/bench_n1.json`): **{b['value']:.1f} outfit-steps/s** ({b['ms_per_step']:.2f} ms per step), "
                   f"`roofline.frac` {rf['frac']:.3f} ({rf['achieved']:.0f} TFLOP/s over {rf['launches_per_step']} GEMM launches per step, "
                   f"{rf['avg_launch_us']:.1f} us per launch from the HIP events)"
                   + (f", `roofline.traffic` {rf['traffic'] / 1e6:.1f} MB per launch against {rf['algorithmic_bytes_per_launch'] / 1e6:.1f} MB algorithmic "
                      f"({rf['traffic'] / rf['algorithmic_bytes_per_launch']:.2f} x)" if rf.get("traffic") else ", `roofline.traffic` null (no PMC summary for these sources)") + ".")
        if kc:
            out.append("* per class, ms per step (launches): " + "; ".join(
                f"{k} {v['ms_per_step']:.2f} ({v['launches_per_step']})" + (f" at {v['tflops']:.0f} TFLOP/s" if v.get("tflops") else "")
                for k, v in kc.items() if v["launches_per_step"]) + ".")
        sec = rf.get("secondary", {})
        if sec:
            out.append("* secondary rooflines: " + "; ".join(f"{k} {v['achieved']:.0f} {v['unit']} = {v['frac']:.3f} of {v['peak']:.0f}" for k, v in sec.items()) + ".")
        cb = b.get("cpu_baseline")
        if cb:
            out.append(f"* CPU baseline beside it: {cb['value']:.4f} {cb['unit']} on {cb['cores']} host threads ({cb['kind']}).")
    f8 = last_json(os.path.join(d, "bench_fp8_n1.json"))
    if f8:
        out.append(f"* `--dtype fp8` (`{tag}/bench_fp8_n1.json`): {f8['value']:.1f} steps/s ({f8['ms_per_step']:.2f} ms per step).")
    tr = last_json(os.path.join(d, "bench_train_n1.json"))
    if tr:
        out.append(f"* training step (`{tag}/bench_train_n1.json`): {tr['value']:.1f} {tr['unit']} ({tr['ms_per_step']:.1f} ms per step).")
    sc = (b or {}).get("secondary_configs")
    if sc:
        parts = []
        for k, v in sc.items():
            if not isinstance(v, dict) or "ms_per_step" not in v:
                continue
            parts.append(f"{k} {v['ms_per_step']:.2f} ms per step" + (f" ({v['steps_per_s']:.1f} steps/s)" if "steps_per_s" in v else "")
                         + (f" ({v['items_per_s']:.1f} items/s)" if "items_per_s" in v else ""))
        if parts:
            out.append(f"* `secondary_configs` of the same default run (`{tag}/bench_n1.json`): " + "; ".join(parts) + ".")
    for name, what in (("bench_sd2base_n1.json", "SD-2-base shape, bf16"), ("bench_sd2base_fp8_n1.json", "SD-2-base shape, fp8"),
                       ("bench_batch64_n1.json", "`--outfits-per-gpu 4` (U-Net batch 64)")):
        x = last_json(os.path.join(d, name))
        if x:
            out.append(f"* {what} (`{tag}/{name}`): {x['value']:.1f} {x['unit']} ({x['ms_per_step']:.2f} ms per step), `roofline.frac` {x['roofline']['frac']:.3f}.")
    va = last_json(os.path.join(d, "bench_vae_n1.json"))
    if va:
        out.append(f"* VAE (`{tag}/bench_vae_n1.json`): {va['value']:.1f} {va['unit']}.")
    for name, what in (("bench_clip_n1.json", "CLIP ViT-L/14 shape (SD-1.5)"), ("bench_clip_sd2base_n1.json", "OpenCLIP ViT-H/14 shape (SD-2)")):
        x = last_json(os.path.join(d, name))
        if x:       # round 6: the prompt table through the HIP CLIP text encoder (bench.py --mode clip)
            out.append(f"* CLIP text encoder, {what} (`{tag}/{name}`): {x['ms_per_step']:.2f} ms per prompt table (51 x 77 tokens), "
                       f"{x['roofline']['achieved']:.1f} TFLOP/s = {x['roofline']['frac']:.3f} of the fp32 MFMA peak"
                       + (f"; CPU oracle beside it {x['cpu_baseline']['value']:.2f} {x['cpu_baseline']['unit']} on {x['cpu_baseline']['cores']} host threads"
                          if x.get("cpu_baseline") else "") + ".")
    st = os.path.join(d, "kernel_stats.csv")
    if os.path.exists(st):
        n, ms, us = gemm_family(st)
        line = f"* rocprofv3 (`{tag}/kernel_stats.csv`): {n} launches of the GEMM family total {ms:.2f} ms = {us:.1f} us per launch"
        if b:
            line += f"; `roofline.avg_launch_us` of the committed bench line (another run, HIP events): {b['roofline']['avg_launch_us']:.1f} us"
        out.append(line + ".")
    st8 = os.path.join(d, "kernel_stats_fp8.csv")
    if os.path.exists(st8):
        n8, ns8 = 0, 0.0
        for r in csv.DictReader(open(st8)):
            if "gemm_fp8_kernel" in r["Name"]:
                n8 += int(r["Calls"]); ns8 += float(r["TotalDurationNs"])
        if n8:
            out.append(f"* rocprofv3 of the fp8 walk (`{tag}/kernel_stats_fp8.csv`): {n8} launches of `gemm_fp8_kernel` total {ns8 / 1e6:.2f} ms = {ns8 / n8 / 1e3:.1f} us per launch.")
    p8 = os.path.join(d, "pmc_traffic_fp8.json")
    if os.path.exists(p8):
        q = json.load(open(p8))
        g8 = q.get("gemm_fp8_kernel")
        out.append(f"* PMC of the fp8 walk (`{tag}/pmc_traffic_fp8.json`): bf16 GEMM family {q['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch over {q['launches']} launches"
                   + (f"; `gemm_fp8_kernel` {g8['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch over {g8['launches']} launches" if g8 else "") + ".")
    pt = os.path.join(d, "pmc_traffic.json")
    if os.path.exists(pt):
        p = json.load(open(pt))
        out.append(f"* PMC (`{tag}/pmc_traffic.json`): FETCH_SIZE {p['fetch_size_kb_per_launch']:.0f} KiB (x2 on gfx950) + WRITE_SIZE "
                   f"{p['write_size_kb_per_launch']:.0f} KiB per launch = {p['hbm_bytes_per_launch'] / 1e6:.1f} MB over {p['launches']} launches.")
    return "\n".join(out) + "\n"


def main():
    tag = sys.argv[1]
    readme = os.path.join(ROOT, "profiles", "README.md")
    s = open(readme).read()
    begin, end = f"<!-- {tag}:generated (scripts/profiles_readme.py {tag}) -->\n", f"<!-- {tag}:end -->\n"
    new = begin + block(tag) + end
    pat = re.compile(re.escape(begin) + ".*?" + re.escape(end), re.S)
    if "--check" in sys.argv:
        m = pat.search(s)
        ok = m is not None and m.group(0) == new
        print("profiles/README.md", tag, "block is", "up to date" if ok else "STALE")
        sys.exit(0 if ok else 1)
    s = pat.sub(lambda _: new, s) if pat.search(s) else s + "\n" + new
    open(readme, "w").write(s)
    print(new)


if __name__ == "__main__":
    main()
