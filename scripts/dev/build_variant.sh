#!/bin/bash
# build the product sources with an extra -D flag into gpurun_ab/<name>/libdifashion_hip.so   usage: build_variant.sh <name> <flags...>
NAME=$1; shift
OUT=/root/repo/gpurun_ab/$NAME; mkdir -p $OUT
cd /root/repo/difashion_amd/csrc
SRCS=$(grep '^SRCS' Makefile | sed 's/SRCS = //')
pids=()
for f in $SRCS; do
  extra=""; [ "$f" = "elementwise.hip" ] && extra="-ffp-contract=off"
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form=1 $extra "$@" -c $f -o $OUT/${f%.hip}.o 2>/dev/null ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 8 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $OUT/*.o -o $OUT/libdifashion_hip.so && rm -f $OUT/*.o && ls -la $OUT/libdifashion_hip.so
