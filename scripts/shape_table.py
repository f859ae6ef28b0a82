#!/usr/bin/env python3
"""Per-shape table from a DFH_PROF_DUMP file: launches grouped by (class, algorithmic flops, algorithmic bytes)."""
import collections
import sys

rows = collections.defaultdict(lambda: [0, 0.0])
for line in open(sys.argv[1]):
    cls, fl, by, ms = line.split()
    k = (cls, float(fl), float(by))
    rows[k][0] += 1; rows[k][1] += float(ms)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tot = sum(v[1] for v in rows.values())
print(f"total {tot / steps:.3f} ms/step")
print(f"{'class':14s} {'GFLOP':>9s} {'MB':>8s} {'n/step':>6s} {'us':>8s} {'ms/step':>8s} {'TF/s':>7s} {'GB/s':>7s}")
for (cls, fl, by), (n, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    us = ms / n * 1e3
    print(f"{cls:14s} {fl / 1e9:9.2f} {by / 1e6:8.2f} {n / steps:6.1f} {us:8.1f} {ms / steps:8.3f} {fl / us / 1e6:7.1f} {by / us / 1e3:7.1f}")
