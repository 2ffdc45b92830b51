#!/bin/bash
# Run ON THE GPU BOX: where the cycles of the pruned scan go, and how long the launch's tail is.
# topk_diag -DM2D_DIAG=8208 (stamps + per-workgroup time line) -> scripts/diag/scan_breakdown.py.  Output: gpurun_out/r06/scan_breakdown.txt
OUT=gpurun_out/${ROUND:-r06}; mkdir -p $OUT/diag
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -Wno-inline-asm -Wno-unused-value"
$CC -DM2D_DIAG=8208 scripts/diag/topk_diag.cpp -o $OUT/diag/topk_8208 2> $OUT/diag/build_topk_8208.log &
$CC -DM2D_DIAG=0 scripts/diag/topk_diag.cpp -o $OUT/diag/topk_0 2> $OUT/diag/build_topk_0.log &
wait
R=$OUT/scan_breakdown.txt; : > $R
export M2D_DIAG_PATTERNS=1
for shape in "65536 100000 4" "65536 1000000 8" "262144 100000 4"; do
  set -- $shape
  for P in 1 0; do
    [ $P = 0 ] && [ $1 != 65536 ] && continue
    echo "== users $1 dishes $2 prune $P" >> $R
    M2D_DIAG_USERS=$1 M2D_DIAG_DISHES=$2 M2D_DIAG_PRUNE=$P M2D_DIAG_REPS=30 timeout -k 5 200 $OUT/diag/topk_0 2>&1 | grep "round 3" >> $R
    M2D_DIAG_USERS=$1 M2D_DIAG_DISHES=$2 M2D_DIAG_PRUNE=$P M2D_DIAG_REPS=30 M2D_DIAG_DUMP=$OUT/diag/scan_dump.bin timeout -k 5 200 $OUT/diag/topk_8208 >> $R 2>&1 || echo FAILED >> $R
    W=$(grep "users per block" $R | tail -1 | sed 's/.*records, \([0-9]*\) users per block/\1/')
    python3 scripts/diag/scan_breakdown.py $OUT/diag/scan_dump.bin $((W / 32)) >> $R 2>&1
    echo progress $shape $P
  done
done
rm -f $OUT/diag/scan_dump.bin
cat $R
