#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per GEMM shape against the known algorithmic bytes (separate --pmc passes, MI355X_MICROARCH.md HBM section).
# Usage (GPU box, repo root): bash scripts/pmc_traffic_calib.sh r02   -> gpurun_out/pmc_calib_r02/summary.txt
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_calib_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 $GRAFT_REPO_ROOT/scripts/pmc_traffic_calib.py $OUT/shapes.json > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 $GRAFT_REPO_ROOT/scripts/pmc_traffic_calib.py > $OUT/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/tcc -o p -- python3 $GRAFT_REPO_ROOT/scripts/pmc_traffic_calib.py > $OUT/tcc.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, json, sys
root = sys.argv[1]
shapes = json.load(open(root + "/shapes.json"))
def per_dispatch(sub):
    rows = {}
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(t in k[:64] for t in ("gemm_bf16_kernel", "gemm_wide_kernel", "gemm_ws_kernel")):
                continue
            rows.setdefault(int(r["Dispatch_Id"]), {"kernel": k})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [rows[i] for i in sorted(rows)]
f, w, t = per_dispatch("fetch"), per_dispatch("write"), per_dispatch("tcc")
n = shapes[0]["launches"]
print(f"{'shape':40s} {'kernel':34s} {'alg read MB':>11s} {'FETCH MB':>9s} {'raw ratio':>9s} {'alg write MB':>12s} {'WRITE MB':>9s} {'ratio':>6s} {'L2 hit':>7s} {'RDREQ*64/FETCH':>14s}")
for i, s in enumerate(shapes):
    g = slice(i * n + 1, (i + 1) * n)          # skip the first launch of each group (cold)
    fs = [x["FETCH_SIZE"] for x in f[g]]; ws = [x["WRITE_SIZE"] for x in w[g]]
    fm = sum(fs) / len(fs) * 1024 / 1e6; wm = sum(ws) / len(ws) * 1024 / 1e6
    tt = t[g]
    hit = sum(x["TCC_HIT_sum"] for x in tt) / max(1.0, sum(x["TCC_HIT_sum"] + x["TCC_MISS_sum"] for x in tt))
    rd = sum(x["TCC_EA0_RDREQ_sum"] for x in tt) / len(tt) * 64 / 1e6
    kn = f[g][0]["kernel"].replace("void (anonymous namespace)::", "")[:34]
    print(f"{s['name']:40s} {kn:34s} {s['read_bytes'] / 1e6:11.1f} {fm:9.1f} {fm / (s['read_bytes'] / 1e6):9.2f} "
          f"{s['write_bytes'] / 1e6:12.1f} {wm:9.1f} {wm / (s['write_bytes'] / 1e6):6.2f} {hit:7.3f} {rd / fm:14.2f}")
PY
