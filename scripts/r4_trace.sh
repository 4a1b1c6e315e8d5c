#!/bin/bash
# kernel traces (rocprofv3 --kernel-trace --stats) of bench variants: scripts/r4_trace.sh <name> [bench args...]
NAME=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r4/trace_$NAME
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --no-cpu --no-e2e --no-secondary --general-steps 0 --no-shard-proxy "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "$NAME rc $?"
for f in $(find "$OUT" -name "*kernel_stats.csv"); do head -16 "$f" | cut -c1-160; done
