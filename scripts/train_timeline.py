#!/usr/bin/env python3
"""Timeline of ONE training step out of a rocprofv3 --kernel-trace CSV (scripts/profile_train.sh): the window between the ends of the last two
adamw_kernel dispatches.  Prints, per kernel name: launches, summed duration, time during which it was the ONLY kernel executing ("alone"),
and time it ran beside another kernel; then the union of all intervals (GPU busy), the idle gaps, and the longest gaps with their neighbours.
   python scripts/train_timeline.py <dir with *kernel_trace.csv> [marker kernel name prefix, default adamw_kernel; sampling: cfg_step_kernel]"""
import collections, csv, glob, os, sys

root = sys.argv[1]
f = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60], r.get("Stream_Id", r.get("Queue_Id", ""))))
rows.sort()
marker = sys.argv[2] if len(sys.argv) > 2 else "adamw_kernel"
ends = [e for s, e, n, q in rows if n.startswith(marker)]
t0, t1 = ends[-2], ends[-1]
win = [(s, e, n, q) for s, e, n, q in rows if s >= t0 and e <= t1]
print(f"step window {(t1 - t0) / 1e6:.2f} ms, {len(win)} dispatches, streams/queues {sorted(set(q for *_, q in win))}")
# sweep line
ev = []
for i, (s, e, n, q) in enumerate(win):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
active = set()
alone = collections.Counter(); shared = collections.Counter(); tot = collections.Counter(); cnt = collections.Counter()
busy = 0; idle = 0; last = t0; gaps = []
prev_name = None
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        if not active:
            idle += dt; gaps.append((dt, last, prev_name))
        else:
            busy += dt
            if len(active) == 1:
                alone[win[next(iter(active))][2]] += dt
            else:
                for j in active: shared[win[j][2]] += dt
    last = t
    if d == 1: active.add(i)
    else:
        active.discard(i); prev_name = win[i][2]
for s, e, n, q in win:
    tot[n] += e - s; cnt[n] += 1
print(f"busy {busy / 1e6:.2f} ms, idle {idle / 1e6:.2f} ms ({len(gaps)} gaps), sum of durations {sum(tot.values()) / 1e6:.2f} ms")
print(f"{'kernel':60s} {'n':>5s} {'sum ms':>8s} {'alone':>8s} {'shared':>8s}")
for n, v in tot.most_common(40):
    print(f"{n:60s} {cnt[n]:5d} {v / 1e6:8.2f} {alone[n] / 1e6:8.2f} {shared[n] / 1e6:8.2f}")
gaps.sort(reverse=True)
print("longest idle gaps (us, after kernel):")
for dt, at, pn in gaps[:12]:
    print(f"  {dt / 1e3:8.1f}  at +{(at - t0) / 1e6:7.2f} ms after {pn}")
hist = collections.Counter()
for dt, *_ in gaps: hist[min(6, int(dt / 1e3) // 2)] += dt
print("idle by gap length (2-us bins, last = >= 12 us):", {f"{2 * k}us": round(v / 1e6, 2) for k, v in sorted(hist.items())})
