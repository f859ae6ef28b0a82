# final artefacts of round 6 (GPU box): full -m gpu suite, rocprofv3 passes, bench lines (after the PMC passes: bench.py reports the traffic
# figure of profiles/rNN/pmc_traffic*.json only when it was measured on these very sources), launch tables, same-box A/B against the round-5 library.
#   bash scripts/dev/r06_final.sh [skip-tests]
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
if [ "${1:-}" != "skip-tests" ]; then
  ( time timeout 2400 python -m pytest tests -m gpu -q --durations=20 ) > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" > $O/status.txt
fi
bash scripts/profile_round.sh r06 > $O/profile_round.log 2>&1
bash scripts/profile_round.sh r06 fp8 > $O/profile_round_fp8.log 2>&1
cp gpurun_out/prof_r06/pmc_traffic.json profiles/r06/pmc_traffic.json
cp gpurun_out/prof_r06_fp8/pmc_traffic_fp8.json profiles/r06/pmc_traffic_fp8.json
python scripts/train_timeline.py gpurun_out/prof_r06/trace cfg_step_kernel > $O/sample_timeline.txt 2>&1
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" >> $O/status.txt
DFH_PROF_TABLE=$O/launch_table.txt timeout 600 python bench.py --no-secondary --no-cpu-baseline > /dev/null 2>&1
timeout 600 python bench.py --dtype fp8 > $O/bench_fp8_n1.json 2> $O/bench_fp8_n1.err
timeout 900 python bench.py --mode train > $O/bench_train_n1.json 2> $O/bench_train_n1.err
timeout 600 python bench.py --config sd2base --no-secondary > $O/bench_sd2base_n1.json 2> $O/bench_sd2base_n1.err
timeout 600 python bench.py --config sd2base --dtype fp8 > $O/bench_sd2base_fp8_n1.json 2>> $O/bench_sd2base_n1.err
timeout 600 python bench.py --outfits-per-gpu 4 --steps 10 --warmup 3 > $O/bench_batch64_n1.json 2> $O/bench_batch64_n1.err
timeout 600 python bench.py --mode vae > $O/bench_vae_n1.json 2> $O/bench_vae_n1.err
timeout 600 python bench.py --mode clip > $O/bench_clip_n1.json 2> $O/bench_clip_n1.err
timeout 600 python bench.py --mode clip --config sd2base --no-cpu-baseline > $O/bench_clip_sd2base_n1.json 2>> $O/bench_clip_n1.err
one() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'])"; }
( for i in 1 2 3; do
    DFH_LIB=$GRAFT_REPO_ROOT/gpurun_ab/r05/libdifashion_hip.so DFH_LIB_ALLOW_ABI_MISMATCH=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-secondary 2>/dev/null | one "sampling round-5 library"
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-secondary 2>/dev/null | one "sampling HEAD           "
  done
  for i in 1 2; do
    DFH_LIB=$GRAFT_REPO_ROOT/gpurun_ab/r05/libdifashion_hip.so DFH_LIB_ALLOW_ABI_MISMATCH=1 python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | one "training round-5 library"
    python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | one "training HEAD           "
  done ) > $O/round_ab_same_box.txt 2>&1
bash scripts/profile_train.sh r06 > $O/profile_train.log 2>&1
python scripts/train_timeline.py gpurun_out/prof_train_r06/trace > $O/train_timeline.txt 2>&1
# keep the summaries, drop the raw traces (the merge back is capped at 64 MiB)
for d in gpurun_out/prof_r06 gpurun_out/prof_r06_fp8 gpurun_out/prof_train_r06; do
  find $d -name "*kernel_stats.csv" -exec cp {} $d/kernel_stats.csv \;
  rm -rf $d/trace $d/pmc_fetch $d/pmc_write $d/pmc_mfma $d/pmc_tcc
done
cat $O/status.txt; tail -3 $O/gpu_tests.log; cut -c1-300 $O/bench_n1.json; cat $O/round_ab_same_box.txt
