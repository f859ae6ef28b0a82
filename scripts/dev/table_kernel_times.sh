cd /tmp && export TMPDIR=/tmp; rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ktrace -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps 2 --warmup 1 --no-profile > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/ktrace -name "*kernel_trace.csv" | head -1); python3 - "$f" <<"PY" > gpurun_out/table_kernel_calls.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r["Kernel_Name"]
    if "table_kernel" in n:
        print(n[:50], r.get("Grid_Size_X") or r.get("Grid_Size"), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
PY
rm -rf gpurun_out/ktrace
