#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small markdown summary."""
import csv, glob, os, sys, collections

root = sys.argv[1]


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
