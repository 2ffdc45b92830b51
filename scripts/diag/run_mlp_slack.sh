#!/bin/bash
# Run ON THE GPU BOX: how much the MLP head's time depends on the slack between its roles.
# m2d_mlp_pc couples all eight waves of a block with ONE s_barrier per period; what slack the gatherers have is the distance
# between a row request and the build that consumes it (two periods, one row register set each).  Measured here, timing-only
# builds of scripts/diag/mlp_diag.cpp (1 M pairs, 200 k users, E = 128, masks grouped), builds alternating on one GPU:
#   stamps        today's kernel with the per-role stamps           (M2D_MLP_DIAG = 2048)
#   ahead1        rows requested ONE period ahead instead of two     (2048 + 16: half the slack)
#   trace/trace1  the barrier-arrival trace of block 0 for both      (65, 81)
#   base on tables that sit in L2 (256 users, 256 dishes): rows that arrive at once = all the slack there could be
# Output: gpurun_out/r06/mlp_slack.txt
OUT=gpurun_out/r06; mkdir -p $OUT/diag
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -Wno-inline-asm -Wno-unused-value"
$CC scripts/diag/mlp_diag.cpp -o $OUT/diag/mlp_base 2> $OUT/diag/b_base.log &
$CC -DM2D_MLP_DIAG=2048 scripts/diag/mlp_diag.cpp -o $OUT/diag/mlp_stamps 2> $OUT/diag/b_2048.log &
$CC -DM2D_MLP_DIAG=2064 scripts/diag/mlp_diag.cpp -o $OUT/diag/mlp_ahead1 2> $OUT/diag/b_2064.log &
$CC -DM2D_MLP_DIAG=65 scripts/diag/mlp_diag.cpp -o $OUT/diag/mlp_trace 2> $OUT/diag/b_65.log &
$CC -DM2D_MLP_DIAG=81 scripts/diag/mlp_diag.cpp -o $OUT/diag/mlp_trace1 2> $OUT/diag/b_81.log &
wait
R=$OUT/mlp_slack.txt; : > $R
for rep in 1 2 3; do
  for B in base stamps ahead1; do
    echo "== $B (grouped masks), pass $rep" >> $R
    timeout -k 5 120 $OUT/diag/mlp_$B 200000 100000 0 0 1 2>&1 | grep -v "^    wave" >> $R || echo FAILED >> $R
  done
  echo progress pass $rep
done
for B in base stamps; do
  echo "== $B, tables that sit in L2 (256 users, 256 dishes)" >> $R
  timeout -k 5 120 $OUT/diag/mlp_$B 256 256 0 0 1 2>&1 | grep -v "^    wave" >> $R || echo FAILED >> $R
done
for B in trace trace1; do
  echo "== $B: block 0, per barrier: interval since the previous release, then each wave's arrival relative to the release" >> $R
  echo "   (waves 0-3 consumers, 4 / 6 gatherers, 5 / 7 loaders; 0 = the last to arrive)" >> $R
  timeout -k 5 120 $OUT/diag/mlp_$B 200000 100000 0 0 1 2>&1 | grep -v "^    wave" >> $R || echo FAILED >> $R
done
tail -5 $R
