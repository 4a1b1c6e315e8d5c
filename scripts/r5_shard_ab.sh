#!/bin/bash
# A/B of a knob on the shard-sized step (unprofiled wall time of 40 steps, 3 repetitions each): usage r5_shard_ab.sh TAG "ENV=1" ["ENV=0" ...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
COMMON="--no-cpu --no-e2e --no-secondary --no-shard-proxy --no-dense-form --general-steps 0 --steps 40 --warmup 5 --queries ${QUERIES:-12500000} ${EXTRA:-}"
for rep in 1 2 3; do
  for v in "" "$@"; do
    name=$(echo "${v:-default}" | tr ' =' '__')
    env $v python3 "$ROOT/bench.py" $COMMON 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', 'rep$rep', d['ms_per_step'], d['value'], d['config'].get('lookup_front'), d['roofline'].get('kernel_ms'))" | tee -a "$OUT/ab.txt"
  done
done
