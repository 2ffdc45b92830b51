#!/bin/bash
# Run ON THE GPU BOX.  Builds and times the ablation variants of the catalogue top-k kernel.
# MASKS="0 16" (M2D_DIAG bit masks); M2D_DIAG_PATTERNS=1 gives the dishes random mask patterns (pruning then has something to do),
# M2D_DIAG_PRUNE=0/1 sets the option.
set -e
mkdir -p gpurun_out/diag
for M in ${MASKS:-0 1 8}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -Wno-inline-asm -DM2D_DIAG=$M scripts/diag/topk_diag.cpp -o gpurun_out/diag/topk_diag_$M 2> gpurun_out/diag/build_$M.log
  timeout -k 5 60 gpurun_out/diag/topk_diag_$M $VARIANT
done
