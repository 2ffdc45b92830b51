#!/bin/bash
# Run ON THE GPU BOX: the pipelined scan's stage wait split into "my DMA pieces have landed" and "every wave is here"
# (-DM2D_DIAG=528: those two in the harness's "body" / "slow path" columns), beside the usual stamps (-DM2D_DIAG=16).
OUT=gpurun_out/r05; mkdir -p $OUT/diag
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -Wno-inline-asm -Wno-unused-value"
for M in ${BUILDS:-16 528}; do $CC -DM2D_DIAG=$M scripts/diag/topk_diag.cpp -o $OUT/diag/topk_$M 2> $OUT/diag/build_topk_$M.log & done; wait
R=$OUT/stage_wait.txt; : > $R
export M2D_DIAG_PATTERNS=1
for M in ${BUILDS:-16 528}; do for P in 0 1; do
  echo "== build $M, prune $P" >> $R
  M2D_DIAG_PRUNE=$P M2D_DIAG_REPS=100 timeout -k 5 120 $OUT/diag/topk_$M >> $R 2>&1 || echo FAILED >> $R
done; done
cat $R
