#!/bin/bash
# rocprofv3 runs of the default bench on the GPU box: kernel trace + stats, then PMC passes (separate runs).
# usage: scripts/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --no-cpu "$@" > "$OUT/bench_trace.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-include-regex "earchKernel|walkKernel|finishKernel|fillNoHitKernel|encodeQueriesKernel|radix_sort|expandHitsKernel|scanTileKernel|scanReduceKernel" --output-format csv -d "$OUT/pmc_$N" -- python3 "$ROOT/bench.py" --no-cpu --steps 2 --warmup 1 "$@" > "$OUT/bench_pmc_$N.log" 2>&1
done
find "$OUT" -name "*.csv" | head -50
for f in $(find "$OUT/trace" -name "*kernel_stats.csv"); do echo "== $f"; head -20 "$f"; done
