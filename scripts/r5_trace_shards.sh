#!/bin/bash
# kernel-trace timelines of the shard-sized steps (what a rank of an 8-GPU strong run does): random, planted, amino
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r5_base}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu --no-e2e --no-secondary --no-shard-proxy --no-dense-form --general-steps 0 --steps 5 --warmup 2"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/random" -- python3 "$ROOT/bench.py" $COMMON --queries 12500000 > "$OUT/random.log" 2>&1
python3 "$ROOT/scripts/trace_gaps.py" "$OUT/random" --around lookupSearch 3 > "$OUT/random_gaps.txt" 2>&1
rocprofv3 --kernel-trace --output-format csv -d "$OUT/planted" -- python3 "$ROOT/bench.py" $COMMON --queries 12500000 --workload planted > "$OUT/planted.log" 2>&1
python3 "$ROOT/scripts/trace_gaps.py" "$OUT/planted" --around orderedSearchKernel 3 > "$OUT/planted_gaps.txt" 2>&1
rocprofv3 --kernel-trace --output-format csv -d "$OUT/amino" -- python3 "$ROOT/bench.py" $COMMON --alphabet amino --queries 6250000 > "$OUT/amino.log" 2>&1
python3 "$ROOT/scripts/trace_gaps.py" "$OUT/amino" --around aminoLookupSearch 3 > "$OUT/amino_gaps.txt" 2>&1
# drop the bulky raw traces, keep the timelines
find "$OUT" -name "*.csv" -size +2M -delete
for f in "$OUT"/*.log; do tail -n 3 "$f" | cut -c1-300; done
