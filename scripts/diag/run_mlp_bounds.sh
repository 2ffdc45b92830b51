#!/bin/bash
# Run ON THE GPU BOX: what bounds m2d_mlp_pc at the current code (timing-only ablations of scripts/diag/mlp_diag.cpp; 1 M pairs,
# 200 k users, E = 128, masks grouped).  Output: gpurun_out/r05/mlp_bounds.txt
OUT=gpurun_out/r05; mkdir -p $OUT/diag
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -Wno-inline-asm -Wno-unused-value"
$CC scripts/diag/mlp_diag.cpp -o $OUT/diag/mlp_base 2> $OUT/diag/b_base.log &
$CC -DM2D_MLP_DIAG=256 scripts/diag/mlp_diag.cpp -o $OUT/diag/mlp_memonly 2> $OUT/diag/b_256.log &
$CC -DM2D_MLP_DIAG=8 scripts/diag/mlp_diag.cpp -o $OUT/diag/mlp_norows 2> $OUT/diag/b_8.log &
$CC -DM2D_MLP_DIAG=12 scripts/diag/mlp_diag.cpp -o $OUT/diag/mlp_norows_nodma 2> $OUT/diag/b_12.log &
$CC -DM2D_MLP_DIAG=4 scripts/diag/mlp_diag.cpp -o $OUT/diag/mlp_nodma 2> $OUT/diag/b_4.log &
$CC -DM2D_MLP_DIAG=2048 scripts/diag/mlp_diag.cpp -o $OUT/diag/mlp_stamped 2> $OUT/diag/b_2048.log &
wait
R=$OUT/mlp_bounds.txt; : > $R
for rep in 1 2; do
  for B in base memonly norows norows_nodma nodma; do
    echo "== $B (grouped masks), pass $rep" >> $R
    timeout -k 5 120 $OUT/diag/mlp_$B 200000 100000 0 0 1 2>&1 | grep -v "^    wave" >> $R || echo FAILED >> $R
  done
done
echo "== stamped" >> $R
timeout -k 5 120 $OUT/diag/mlp_stamped 200000 100000 0 0 1 >> $R 2>&1
echo "== base, tables that sit in L2 (256 users, 256 dishes)" >> $R
timeout -k 5 120 $OUT/diag/mlp_base 256 256 0 0 1 2>&1 | grep -v "^    wave" >> $R
cat $R
