cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/run31_smoke.txt 2>&1; echo "smoke rc=$?"
tail -3 gpurun_out/r05/run31_smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/run31_bench.json 2> gpurun_out/r05/run31_bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r05/run31_bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], list(d['secondary_configs'].keys()), d['cpu_baseline']['value'])"
DFH_LIB=scripts/probes/build/libdifashion_probes.so timeout 900 python -m pytest scripts/probes/tests -x -q 2>&1 | tail -2
