#!/bin/bash
# Run ON THE GPU BOX.  A/B of the bf16 MFMA shape on the two clock-limited kernels, timing-only builds (wrong results):
# v_mfma_f32_32x32x16_bf16 (what ships) against v_mfma_f32_16x16x32_bf16 on the same accumulator registers, same LDS reads, same
# vector work (-DM2D_DIAG=256 / -DM2D_MLP_SHAPE16=1).  Wall time from unstamped builds, in-kernel clock (s_memtime / s_memrealtime)
# from stamped ones; the two shapes alternate on one GPU.  Output: gpurun_out/r05/shape_ab.txt
# The MLP half (-DM2D_MLP_SHAPE16) exists at commit 64ea750 only: the consumers have run on 16x16x32 for real since 94a2c20
# (profiles/r05_mfma_shape_ab.txt keeps that commit's record); at today's code ONLY=retrieval is the part that still means something.
set -o pipefail
OUT=gpurun_out/r05; mkdir -p $OUT/diag
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -Wno-inline-asm -Wno-unused-value"
ONLY=${ONLY:-retrieval}   # retrieval | mlp | both (mlp: at commit 64ea750)
if [ $ONLY != mlp ]; then for M in 0 256 16 272; do $CC -DM2D_DIAG=$M scripts/diag/topk_diag.cpp -o $OUT/diag/topk_$M 2> $OUT/diag/build_topk_$M.log & done; fi
$CC -DM2D_MLP_SHAPE16=0 scripts/diag/mlp_diag.cpp -o $OUT/diag/mlp_s0 2> $OUT/diag/build_mlp_s0.log &
$CC -DM2D_MLP_SHAPE16=1 scripts/diag/mlp_diag.cpp -o $OUT/diag/mlp_s1 2> $OUT/diag/build_mlp_s1.log &
$CC -DM2D_MLP_SHAPE16=0 -DM2D_MLP_DIAG=2048 scripts/diag/mlp_diag.cpp -o $OUT/diag/mlp_s0_clk 2> $OUT/diag/build_mlp_s0c.log &
$CC -DM2D_MLP_SHAPE16=1 -DM2D_MLP_DIAG=2048 scripts/diag/mlp_diag.cpp -o $OUT/diag/mlp_s1_clk 2> $OUT/diag/build_mlp_s1c.log &
wait
ls -la $OUT/diag | grep -v log
R=$OUT/shape_ab_$ONLY.txt; : > $R
export M2D_DIAG_PATTERNS=1
for rep in 1 2 3; do
  [ $ONLY = mlp ] && break
  for M in 0 256; do
    echo "== retrieval, every tile (prune 0), shape build $M, pass $rep" >> $R
    M2D_DIAG_PRUNE=0 M2D_DIAG_REPS=150 timeout -k 5 120 $OUT/diag/topk_$M >> $R 2>&1 || echo FAILED >> $R
    echo "== retrieval, pruned, shape build $M, pass $rep" >> $R
    M2D_DIAG_PRUNE=1 M2D_DIAG_REPS=400 timeout -k 5 120 $OUT/diag/topk_$M >> $R 2>&1 || echo FAILED >> $R
  done
  echo progress retrieval pass $rep
done
for M in 16 272; do
  [ $ONLY = mlp ] && break
  echo "== retrieval STAMPED (clock), every tile, build $M" >> $R
  M2D_DIAG_PRUNE=0 M2D_DIAG_REPS=100 timeout -k 5 120 $OUT/diag/topk_$M >> $R 2>&1 || echo FAILED >> $R
  echo "== retrieval STAMPED (clock), pruned, build $M" >> $R
  M2D_DIAG_PRUNE=1 M2D_DIAG_REPS=300 timeout -k 5 120 $OUT/diag/topk_$M >> $R 2>&1 || echo FAILED >> $R
done
echo progress retrieval stamped
[ $ONLY = retrieval ] && exit 0
for rep in 1 2 3; do
  for S in s0 s1; do
    echo "== MLP head (1 M pairs, 200 k users, E = 128, masks grouped), build $S, pass $rep" >> $R
    timeout -k 5 120 $OUT/diag/mlp_$S 200000 100000 0 0 1 >> $R 2>&1 || echo FAILED >> $R
    echo "== MLP head, all-ones masks (every k-block), build $S, pass $rep" >> $R
    timeout -k 5 120 $OUT/diag/mlp_$S 200000 100000 0 0 0 >> $R 2>&1 || echo FAILED >> $R
  done
  echo progress mlp pass $rep
done
for S in s0_clk s1_clk; do
  echo "== MLP head STAMPED (clock), build $S" >> $R
  timeout -k 5 120 $OUT/diag/mlp_$S 200000 100000 0 0 1 >> $R 2>&1 || echo FAILED >> $R
done
grep -c . $R
