# rocprofv3 --kernel-trace --stats of the training step (bench.py --mode train), per-kernel table per STEP into gpurun_out/train_kstats.txt
cd /tmp && export TMPDIR=/tmp
STEPS=3
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ktrace -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps $STEPS --warmup 1 --no-profile > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/ktrace -name "*kernel_trace.csv" | head -1)
python3 - "$f" $STEPS <<"PY" > gpurun_out/train_kstats.txt
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1]))); steps=int(sys.argv[2])
# keep the launches of the last `steps` optimizer steps: everything after the (steps+1)-th last adamw launch
idx=[i for i,r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
start=idx[-steps-1]+1 if len(idx)>steps else 0
agg=collections.defaultdict(lambda:[0,0.0])
for r in rows[start:idx[-1]+1]:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")
    n=n.split("(")[0] if not n.startswith("gemm") and "kernel<" not in n else n.split("(")[0]
    a=agg[n]; a[0]+=1; a[1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
tot=sum(v[1] for v in agg.values())
print(f"{'kernel':70s} {'n/step':>7s} {'avg us':>9s} {'ms/step':>8s} {'%':>5s}")
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    print(f"{k[:70]:70s} {v[0]/steps:7.1f} {v[1]/v[0]:9.1f} {v[1]/steps/1e3:8.3f} {100*v[1]/tot:5.1f}")
print("total kernel ms/step", tot/steps/1e3)
PY
rm -rf gpurun_out/ktrace
