#!/bin/bash
# Run ON THE GPU BOX: diagnostic builds of the retrieval harness (scripts/diag/topk_diag.cpp), every tile and pruned.
#   BUILDS="16 528"  stamps; 528 = the stage wait split into "my DMA pieces have landed" / "every wave is here" (body / slow path columns)
#   BUILDS="0 8 1024" timing-only ablations: 8 = no candidate handling, 1024 = a step cut down to barrier + DMA + body
#   BUILDS="4112"    time line of workgroup (0, 0)'s steps, two SIMD-mates side by side
#   M2D_DIAG_DISHES / M2D_DIAG_USERS: the shape (default 65 536 users x 100 000 dishes)
OUT=gpurun_out/r05; mkdir -p $OUT/diag
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -Wno-inline-asm -Wno-unused-value"
for M in ${BUILDS:-16 528}; do $CC -DM2D_DIAG=$M scripts/diag/topk_diag.cpp -o $OUT/diag/topk_$M 2> $OUT/diag/build_topk_$M.log & done; wait
R=$OUT/topk_diag.txt; : > $R
export M2D_DIAG_PATTERNS=1
for M in ${BUILDS:-16 528}; do for P in 0 1; do
  echo "== build $M, prune $P" >> $R
  M2D_DIAG_PRUNE=$P M2D_DIAG_REPS=100 timeout -k 5 120 $OUT/diag/topk_$M >> $R 2>&1 || echo FAILED >> $R
done; done
cat $R
