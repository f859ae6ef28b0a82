#!/bin/bash
# PMC passes on one GEMM shape.  Usage: bash scripts/pmc_gemm.sh conv320
set -u
W=${1:-conv320}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/$tag -o p -- python3 $GRAFT_REPO_ROOT/scripts/gemm_one.py $W > $OUT/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_bf16" not in k: continue
        agg[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k[:60], r["Counter_Name"])] += 1
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:36s} {x / cnt[(k, c)]:.4g} / dispatch")
PY
