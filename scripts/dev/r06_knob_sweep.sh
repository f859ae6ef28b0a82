# round 6: re-check the walk's tuning knobs on the final sources, one box, alternating with the default (each knob was tuned in the round it was added)
O=gpurun_out/r06_knobs; mkdir -p $O
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-28s %.3f' % ('$*', d['ms_per_step']))"; }
for k in X=0 DFH_GEMM_BIG=1 DFH_GEMM_BIG=3 DFH_GEMM_BIGG=0 X=0 DFH_DEEP4=0 DFH_DEEP4=1 DFH_WINO=0 DFH_WINO=1 DFH_WINO_MAXHW=1024 X=0 DFH_GN_FOLD=0 DFH_GN_FOLD=640 DFH_GSTAT128=0 DFH_MLP_FUSED=0 DFH_CFG_DEDUP=0 X=0 DFH_BATCH_BIG=0 DFH_BATCH_NMAJOR=0 DFH_W_BLOCKED=0 DFH_QKV_MERGE=0 DFH_FFP_FOLD=0 DFH_UPS_PHASE=0 DFH_ATTN_XS=0 X=0; do run $k; done | tee $O/knob_sweep.txt
