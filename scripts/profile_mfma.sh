#!/bin/bash
# Run ON THE GPU BOX: kernel-trace + PMC passes for the MFMA kernels.  Usage: scripts/profile_mfma.sh <tag>
set -e -o pipefail
TAG=${1:-r01_mfma}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 scripts/prof_mfma.py both $OUT/mfma_bench.json > $OUT/stats.log 2>&1
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
         "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 scripts/prof_mfma.py > $OUT/pmc_$N.log 2>&1 || echo "pass $N failed"
done
python3 scripts/summarize_rocprof.py $OUT $TAG > $OUT/summary_print.log 2>&1 || true
python3 - $OUT <<'PY'
import json,sys
s=json.load(open(sys.argv[1]+"/summary.json"))
for k,v in s["kernels"].items():
    if "m2d" in k: print(k[:70], {a:round(b,1) for a,b in v.items()})
for k,v in s["counters"].items():
    if "m2d_topk" in k or "m2d_mlp" in k:
        print(k[:60]); print({a:(b["avg_per_dispatch"] if isinstance(b,dict) else b) for a,b in v.items()})
PY
