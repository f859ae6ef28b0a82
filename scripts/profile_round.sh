#!/bin/bash
# rocprofv3 passes for profiles/: kernel trace + stats, then PMC (HBM bytes) in their own runs.
# Usage (on the GPU box, from the repo root):  bash scripts/profile_round.sh r01 [fp8]
#   second argument "fp8": the same passes over `bench.py --dtype fp8` (BASELINE configs[4]) -> gpurun_out/prof_<tag>_fp8, whose
#   summary writes pmc_traffic_fp8.json (the traffic figure bench.py reports for --dtype fp8 comes from THAT file or is null)
set -u
TAG=${1:-r01}
DT=${2:-bf16}
SUF=""; EXTRA=""
if [ "$DT" = "fp8" ]; then SUF="_fp8"; EXTRA="--dtype fp8"; fi
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG$SUF
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-secondary $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o $TAG -- $BENCH > $OUT/pmc_mfma.log 2>&1
# round 6: L2 (TCC) hit rate and fabric-side read requests per kernel -- backs "the re-reads beyond the algorithmic bytes are L2 / MALL hits"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/pmc_tcc -o $TAG -- $BENCH > $OUT/pmc_tcc.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT -name "*.csv" | head -30
python3 scripts/summarize_profile.py $OUT $DT > $OUT/summary.md 2>&1
cat $OUT/summary.md | head -60
