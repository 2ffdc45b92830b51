#!/bin/bash
# Run ON THE GPU BOX: instruction-fetch / scalar counters of the every-tile pipelined scan, for a few diagnostic builds
# (BUILDS: -DM2D_DIAG masks; binaries from run_stage_wait.sh are rebuilt here).  Output: gpurun_out/r05/scan_ifetch.txt
R=$PWD; OUT=$R/gpurun_out/r05; mkdir -p $OUT/diag
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -Wno-inline-asm -Wno-unused-value"
for M in ${BUILDS:-0 8 1024}; do $CC -DM2D_DIAG=$M scripts/diag/topk_diag.cpp -o $OUT/diag/topk_$M 2> $OUT/diag/build_topk_$M.log & done; wait
cd /tmp && export TMPDIR=/tmp
export M2D_DIAG_PATTERNS=1 M2D_DIAG_PRUNE=${PRUNE:-0}
: > $OUT/scan_ifetch.txt
for M in ${BUILDS:-0 8 1024}; do
  i=0
  for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INSTS_LDS"; do
    i=$((i+1)); rm -rf /tmp/pmc_$M_$i
    timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_${M}_$i -o run --output-format csv -- $OUT/diag/topk_$M > /tmp/pmc_${M}_$i.log 2>&1 || echo "pmc pass $i of build $M failed" >> $OUT/scan_ifetch.txt
    python3 - $M /tmp/pmc_${M}_$i >> $OUT/scan_ifetch.txt <<'PY'
import csv, glob, sys, collections
m, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'bf16_pipe2' not in row['Kernel_Name']: continue
        a = acc[row['Counter_Name']]; a[0] += float(row['Counter_Value']); a[1] += 1
for k, (s, n) in sorted(acc.items()): print('build %s  %-28s %.4e per dispatch (%d dispatches)' % (m, k, s / n, n))
PY
  done
done
cat $OUT/scan_ifetch.txt
