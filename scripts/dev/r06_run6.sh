# round 6, GPU run 6: attention ceiling evidence (PMC + in-kernel phase stamps), rocprofv3 passes of the round (bf16 + fp8 walks)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_run6; mkdir -p $O
cd $GRAFT_REPO_ROOT
bash scripts/pmc_attn.sh r06_run6/pmc_attn > $O/pmc_attn.txt 2>&1; tail -30 $O/pmc_attn.txt
( export DFH_LIB=$GRAFT_REPO_ROOT/scripts/probes/build/libdifashion_probes.so DFH_ATTN_VARIANT=9; python scripts/attn_one.py ) > $O/attn_stamps.txt 2>&1; grep "attn prof\|self" $O/attn_stamps.txt
python scripts/attn_microbench.py > $O/attn_microbench.txt 2>&1; grep -v amdgpu $O/attn_microbench.txt | tail -12
bash scripts/profile_round.sh r06 > $O/profile_round.log 2>&1; tail -40 $O/profile_round.log
bash scripts/profile_round.sh r06 fp8 > $O/profile_round_fp8.log 2>&1; tail -5 $O/profile_round_fp8.log
rm -rf gpurun_out/r06_run6/pmc_attn/*/  # raw counter dirs
