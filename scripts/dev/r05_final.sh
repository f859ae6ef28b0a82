# final artefacts of the round (GPU box): full -m gpu suite, rocprofv3 passes, bench lines (after the PMC passes: bench.py reports the traffic
# figure of profiles/rNN/pmc_traffic*.json only when it was measured on these very sources), launch tables, same-box round A/B.
#   bash scripts/dev/r05_final.sh [skip-tests]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05f
O=gpurun_out/r05f
if [ "${1:-}" != "skip-tests" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" > $O/status.txt
fi
bash scripts/profile_round.sh r05 > $O/profile_round.log 2>&1
bash scripts/profile_round.sh r05 fp8 > $O/profile_round_fp8.log 2>&1
cp gpurun_out/prof_r05/pmc_traffic.json profiles/r05/pmc_traffic.json
cp gpurun_out/prof_r05_fp8/pmc_traffic_fp8.json profiles/r05/pmc_traffic_fp8.json
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" >> $O/status.txt
DFH_PROF_TABLE=$O/launch_table.txt timeout 600 python bench.py --no-secondary --no-cpu-baseline > /dev/null 2>&1
timeout 600 python bench.py --dtype fp8 > $O/bench_fp8_n1.json 2> $O/bench_fp8_n1.err
DFH_PROF_TABLE=$O/launch_table_fp8.txt timeout 600 python bench.py --dtype fp8 --no-cpu-baseline > /dev/null 2>&1
timeout 900 python bench.py --mode train > $O/bench_train_n1.json 2> $O/bench_train_n1.err
DFH_PROF_TABLE=$O/launch_table_train.txt timeout 600 python bench.py --mode train --steps 4 --warmup 2 > /dev/null 2>&1
timeout 600 python bench.py --config sd2base --no-secondary > $O/bench_sd2base_n1.json 2> $O/bench_sd2base_n1.err
timeout 600 python bench.py --config sd2base --dtype fp8 > $O/bench_sd2base_fp8_n1.json 2>> $O/bench_sd2base_n1.err
timeout 600 python bench.py --outfits-per-gpu 4 --steps 10 --warmup 3 > $O/bench_batch64_n1.json 2> $O/bench_batch64_n1.err
timeout 600 python bench.py --mode vae > $O/bench_vae_n1.json 2> $O/bench_vae_n1.err
bash scripts/dev/round_ab.sh > $O/round_ab_same_box.txt 2>&1
bash scripts/profile_train.sh r05 > $O/profile_train.log 2>&1
python scripts/train_timeline.py gpurun_out/prof_train_r05/trace > $O/train_timeline.txt 2>&1
# keep the summaries, drop the raw traces (the merge back is capped at 64 MiB)
for d in gpurun_out/prof_r05 gpurun_out/prof_r05_fp8 gpurun_out/prof_train_r05; do
  find $d -name "*kernel_stats.csv" -exec cp {} $d/kernel_stats.csv \;
  rm -rf $d/trace $d/pmc_fetch $d/pmc_write $d/pmc_mfma
done
cat $O/status.txt; tail -3 $O/gpu_tests.log; cut -c1-300 $O/bench_n1.json; cat $O/round_ab_same_box.txt
