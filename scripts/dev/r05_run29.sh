cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f
mkdir -p $O
timeout 600 python bench.py --config sd2base --no-secondary > $O/bench_sd2base_n1.json 2> $O/bench_sd2base_n1.err
timeout 600 python bench.py --config sd2base --dtype fp8 > $O/bench_sd2base_fp8_n1.json 2>> $O/bench_sd2base_n1.err
timeout 600 python bench.py --outfits-per-gpu 4 --steps 10 --warmup 3 > $O/bench_batch64_n1.json 2> $O/bench_batch64_n1.err
timeout 600 python bench.py --no-secondary --no-cpu-baseline > $O/bench_sd15_same_box.json 2>/dev/null
timeout 600 python bench.py --dtype fp8 --no-cpu-baseline > $O/bench_sd15_fp8_same_box.json 2>/dev/null
echo "one box, one gpurun call (boxes of the pool differ by up to 8 %: compare within this file)" > $O/other_workloads_same_box.txt
for f in bench_sd15_same_box bench_sd15_fp8_same_box bench_sd2base_n1 bench_sd2base_fp8_n1 bench_batch64_n1; do python -c "import json; d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); print('%-26s %8.3f %s  %8.3f ms per step   roofline.frac %.3f' % ('$f', d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac']))" >> $O/other_workloads_same_box.txt; done
cat $O/other_workloads_same_box.txt
