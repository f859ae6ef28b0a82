#!/bin/bash
# rocprofv3 kernel trace + stats of the training-step bench (BASELINE configs[2]).  Usage: bash scripts/profile_train.sh r01
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_train_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps 3 --warmup 1 --no-profile > $OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
ST=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$ST" <<'PY' > $OUT/kernel_stats_train.md
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for r in rows[:30]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:80]
    print(f"| {n} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
PY
cp "$ST" $OUT/kernel_stats_train.csv
cat $OUT/kernel_stats_train.md
tail -2 $OUT/trace.log
