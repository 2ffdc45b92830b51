#!/bin/bash
# Run ON THE GPU BOX: time line (s_memtime) of workgroup (0, 0)'s steps in the pipelined scan, SIMD-mates side by side
# (-DM2D_DIAG=4112).  Output: gpurun_out/r05/scan_timeline.txt
OUT=gpurun_out/r05; mkdir -p $OUT/diag
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -Wno-inline-asm -Wno-unused-value"
$CC -DM2D_DIAG=4112 scripts/diag/topk_diag.cpp -o $OUT/diag/topk_4112 2> $OUT/diag/build_topk_4112.log
export M2D_DIAG_PATTERNS=1
M2D_DIAG_PRUNE=${PRUNE:-0} timeout -k 5 120 $OUT/diag/topk_4112 > $OUT/scan_timeline.txt 2>&1
cat $OUT/scan_timeline.txt
