# round 6, GPU run 4: PMC of the persistent probe kernel vs the tile kernel (one counter group per run); probe-library parity tests
O=$GRAFT_REPO_ROOT/gpurun_out/r06_run4; mkdir -p $O
export DFH_LIB=$GRAFT_REPO_ROOT/scripts/probes/build/libdifashion_probes.so DFH_LIB_ALLOW_ABI_MISMATCH=0
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest scripts/probes/tests -q -x -k "persistent" 2>&1 | tail -5 > $O/probe_tests.log; tail -3 $O/probe_tests.log
cd /tmp && export TMPDIR=/tmp
for W in lin320k320_tile lin320k320_persist; do
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $grp --output-format csv -d $O/$W/$tag -o p -- python3 $GRAFT_REPO_ROOT/scripts/gemm_one.py $W > $O/$W.$tag.log 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python3 - $O <<'PY' | tee $O/pmc_persist_vs_tile.txt
import csv, glob, sys, collections
root = sys.argv[1]
for W in ("lin320k320_tile", "lin320k320_persist"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(root + "/" + W + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_bf16" not in k and "gemm_persist" not in k: continue
            agg[k[:70]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k[:70], r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(W, k)
        for c, x in sorted(v.items()):
            print(f"   {c:36s} {x / cnt[(k, c)]:.4g} / dispatch")
PY
rm -rf $O/lin320k320_tile $O/lin320k320_persist
