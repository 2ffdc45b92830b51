#!/bin/bash
# Run ON THE GPU BOX.  A/B of the bf16 MFMA shape on the two clock-limited kernels, timing-only builds (wrong results):
# v_mfma_f32_32x32x16_bf16 (what ships) against v_mfma_f32_16x16x32_bf16 on the same accumulator registers, same LDS reads, same
# vector work (-DM2D_DIAG=256 / -DM2D_MLP_SHAPE16=1).  Wall time from unstamped builds, in-kernel clock (s_memtime / s_memrealtime)
# from stamped ones; the two shapes alternate on one GPU.  Output: gpurun_out/r05/shape_ab.txt
# The MLP half of round 5's A/B is gone with the macro it switched (M2D_MLP_SHAPE16; the consumers run 16x16x32 unconditionally
# since 94a2c20): profiles/r05_mfma_shape_ab.txt keeps that record.  This script A/Bs the retrieval kernel only.
set -o pipefail
OUT=gpurun_out/${ROUND:-r06}; mkdir -p $OUT/diag
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -Wno-inline-asm -Wno-unused-value"
for M in 0 256 16 272; do $CC -DM2D_DIAG=$M scripts/diag/topk_diag.cpp -o $OUT/diag/topk_$M 2> $OUT/diag/build_topk_$M.log & done
wait
ls -la $OUT/diag | grep -v log
R=$OUT/shape_ab_retrieval.txt; : > $R
export M2D_DIAG_PATTERNS=1
for rep in 1 2 3; do
  for M in 0 256; do
    echo "== retrieval, every tile (prune 0), shape build $M, pass $rep" >> $R
    M2D_DIAG_PRUNE=0 M2D_DIAG_REPS=150 timeout -k 5 120 $OUT/diag/topk_$M >> $R 2>&1 || echo FAILED >> $R
    echo "== retrieval, pruned, shape build $M, pass $rep" >> $R
    M2D_DIAG_PRUNE=1 M2D_DIAG_REPS=400 timeout -k 5 120 $OUT/diag/topk_$M >> $R 2>&1 || echo FAILED >> $R
  done
  echo progress retrieval pass $rep
done
for M in 16 272; do
  echo "== retrieval STAMPED (clock), every tile, build $M" >> $R
  M2D_DIAG_PRUNE=0 M2D_DIAG_REPS=100 timeout -k 5 120 $OUT/diag/topk_$M >> $R 2>&1 || echo FAILED >> $R
  echo "== retrieval STAMPED (clock), pruned, build $M" >> $R
  M2D_DIAG_PRUNE=1 M2D_DIAG_REPS=300 timeout -k 5 120 $OUT/diag/topk_$M >> $R 2>&1 || echo FAILED >> $R
done
cat $R
