#!/bin/bash
# PMC passes on the 64x64-level attention launch.  Usage: bash scripts/pmc_attn.sh
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_attn}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/$tag -o p -- python3 $GRAFT_REPO_ROOT/scripts/attn_one.py > $OUT/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attention_" not in k: continue
        agg[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k[:60], r["Counter_Name"])] += 1
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:36s} {x / cnt[(k, c)]:.4g} / dispatch")
PY
