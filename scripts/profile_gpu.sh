#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats pass + separate PMC passes of the same bench command.
# Usage: scripts/profile_gpu.sh <tag> [bench args...]      -> gpurun_out/prof_<tag>/...
set -e -o pipefail
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH_ARGS="--steps 50 --warmup 5 --no-cpu-baseline --no-side $@"
echo "[profile] stats pass"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $BENCH_ARGS > $OUT/stats_bench.json 2> $OUT/stats.log
# STATS_ONLY=1: the kernel-trace pass alone (e.g. `STATS_ONLY=1 scripts/profile_gpu.sh r04_cfg2_noskip --opt skip_masked=0`:
# the SURVEY 8d-literal leg, every row of the user block fetched -- its average duration is what roofline.survey_8d_ms times)
if [ -n "$STATS_ONLY" ]; then python3 scripts/summarize_rocprof.py $OUT $TAG; exit 0; fi
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  N=$(echo $C | tr ' ' '_')
  echo "[profile] pmc pass $C"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 bench.py $BENCH_ARGS > $OUT/pmc_${N}_bench.json 2> $OUT/pmc_$N.log
done
python3 scripts/summarize_rocprof.py $OUT $TAG
