#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small markdown summary."""
import csv, glob, os, sys, collections

root = sys.argv[1]
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"      # "fp8": the walk of bench.py --dtype fp8 (pmc_traffic_fp8.json)


def find(sub, pat):
    g = glob.glob(os.path.join(root, sub, "**", pat), recursive=True)
    return g[0] if g else None


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name[:70]


print(f"# rocprofv3 summary ({os.path.basename(root)})\n")
st = find("trace", "*kernel_stats.csv")
if st:
    print("## kernel stats (rocprofv3 --kernel-trace --stats; bench.py --steps 4 --warmup 2 => 6 steps + setup)\n")
    print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
    rows = list(csv.DictReader(open(st)))
    for r in rows[:18]:
        print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("pmc_mfma", None)):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    print(f"\n## PMC pass {sub}\n")
    names = sorted({c for v in agg.values() for c in v})
    print("| kernel | dispatches | " + " | ".join(f"{n} (sum)" for n in names) + " | " + " | ".join(f"{n} / dispatch" for n in names) + " |")
    print("|---|---|" + "---|" * (2 * len(names)))
    top = sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:10]
    for k, v in top:
        n = max(cnt[(k, c)] for c in names if (k, c) in cnt)
        print(f"| {k} | {n} | " + " | ".join(f"{v.get(c, 0):.4g}" for c in names) + " | " + " | ".join(f"{v.get(c, 0) / n:.4g}" for c in names) + " |")

# ---- machine-readable HBM traffic per launch of the dominant kernel (bench.py "roofline.traffic")
import json
def per_kernel(sub):
    f = find(sub, "*counter_collection.csv")
    tot, n = collections.Counter(), collections.Counter()
    if f:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]); tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n
ft, fn = per_kernel("pmc_fetch")
wt, wn = per_kernel("pmc_write")
# the dominant family as bench.py defines it: every conv3x3 / 1x1 / linear launch of the bf16 classes (the fused feed-forward kernel included)
gem = [k for k in ft if k.startswith(("gemm_bf16_kernel", "gemm_wide_kernel", "mlp2_fused_kernel"))]
if gem:
    launches = sum(fn[k] for k in gem)
    fetch_kb = sum(ft[k] for k in gem); write_kb = sum(wt.get(k, 0) for k in gem)
    out = dict(kernel="gemm_bf16_kernel + gemm_wide_kernel + mlp2_fused_kernel (all tile variants)", dtype=dtype, launches=launches,
               fetch_size_kb_per_launch=fetch_kb / launches, write_size_kb_per_launch=write_kb / max(1, sum(wn.get(k, 0) for k in gem)),
               correction="gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> x2 (MI355X_MICROARCH.md HBM section); WRITE_SIZE uncorrected",
               hbm_bytes_per_launch=(2 * fetch_kb / launches + write_kb / max(1, sum(wn.get(k, 0) for k in gem))) * 1024)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        import bench
        out["kernel_source_hash"] = bench.kernel_source_hash()      # bench.py reports this summary only for these exact sources
    except Exception as e:
        out["kernel_source_hash"] = None
    # fp8 walk: the e4m3 class beside it, from the same passes (gemm_fp8_kernel launches)
    f8 = [k for k in ft if k.startswith("gemm_fp8_kernel")]
    if f8:
        n8 = sum(fn[k] for k in f8)
        out["gemm_fp8_kernel"] = dict(launches=n8, fetch_size_kb_per_launch=sum(ft[k] for k in f8) / n8,
                                      write_size_kb_per_launch=sum(wt.get(k, 0) for k in f8) / max(1, sum(wn.get(k, 0) for k in f8)),
                                      hbm_bytes_per_launch=(2 * sum(ft[k] for k in f8) / n8 + sum(wt.get(k, 0) for k in f8) / max(1, sum(wn.get(k, 0) for k in f8))) * 1024)
    json.dump(out, open(os.path.join(root, "pmc_traffic.json" if dtype == "bf16" else f"pmc_traffic_{dtype}.json"), "w"), indent=1)

# ---- derived per-kernel figures: MFMA-busy %, achieved HBM GB/s (FETCH x2 + WRITE over the traced duration)
mf = find("pmc_mfma", "*counter_collection.csv")
dur = {}
if st:
    for r in csv.DictReader(open(st)):
        dur[short(r["Name"])] = (float(r["TotalDurationNs"]), int(r["Calls"]))
if mf:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    fnm = collections.Counter()
    for r in csv.DictReader(open(mf)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            fnm[short(r["Kernel_Name"])] += 1
    print("\n## derived per kernel\n")
    print("MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES): SQ_BUSY_CYCLES is summed over the 32 shader engines, each with")
    print("32 SIMDs (8 CUs x 4), so the denominator is the SIMD-cycles of the launch; HBM GB/s = (FETCH_SIZE x 2 + WRITE_SIZE) KiB per")
    print("dispatch over the traced average duration (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section).\n")
    tc = find("pmc_tcc", "*counter_collection.csv")
    tcc = collections.defaultdict(lambda: collections.defaultdict(float))
    tcn = collections.Counter()
    if tc:
        for r in csv.DictReader(open(tc)):
            tcc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "TCC_HIT_sum":
                tcn[short(r["Kernel_Name"])] += 1
    print("clock MHz = GRBM_GUI_ACTIVE per dispatch / 8 XCDs / the traced average duration (two different passes: +-10 %); L2 hit % =")
    print("TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum); fabric reads MB = TCC_EA0_RDREQ_sum x 128 B per dispatch (wide reads: the request the")
    print("counter tallies at 64 B is a 128-B one, MI355X_MICROARCH.md HBM section) -- reads that left the XCD's L2 for MALL / HBM.\n")
    print("| kernel | MFMA busy % | HBM GB/s | avg us | clock MHz | L2 hit % | fabric reads MB / launch |\n|---|---|---|---|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:16]:
        sqb = v.get("SQ_BUSY_CYCLES", 0)
        busy = 100.0 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (sqb * 32) if sqb else 0.0
        gbs = ""
        if k in dur and dur[k][0] > 0 and k in ft:
            byts = (2 * ft[k] / max(1, fn[k]) + wt.get(k, 0) / max(1, wn.get(k, 1))) * 1024 * dur[k][1]
            gbs = f"{byts / dur[k][0]:.0f}"
        avg_us = dur[k][0] / dur[k][1] / 1e3 if k in dur else 0.0
        mhz = ""
        if avg_us > 0 and fnm.get(k):
            mhz = f"{v.get('GRBM_GUI_ACTIVE', 0) / fnm[k] / 8.0 / avg_us:.0f}"
        hit, fab = "", ""
        if k in tcc and tcn[k]:
            h, m = tcc[k].get("TCC_HIT_sum", 0.0), tcc[k].get("TCC_MISS_sum", 0.0)
            hit = f"{100.0 * h / (h + m):.1f}" if h + m > 0 else ""
            fab = f"{tcc[k].get('TCC_EA0_RDREQ_sum', 0.0) * 128.0 / tcn[k] / 1e6:.1f}"
        print(f"| {k} | {busy:.1f} | {gbs} | {avg_us:.1f} | {mhz} | {hit} | {fab} |")
