#!/bin/bash
# amino 2e9 / 2e8: lookup forced, general forced, default; planted too (--workload planted)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r5_amino}; mkdir -p "$OUT"
COMMON="--alphabet amino --no-cpu --no-e2e --no-shard-proxy --no-dense-form --steps 5 --warmup 2"
for n in 2e9 2e8; do
  for w in random planted; do
    for v in "" "AWFM_GPU_AMINO_LOOKUP=1" "AWFM_GPU_AMINO_LOOKUP=0"; do
      name=$(echo "n${n}_${w}_${v:-default}" | tr ' =' '__')
      env $v python3 "$ROOT/bench.py" $COMMON --text-len $n --workload $w 2>"$OUT/$name.err" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('$name', d['ms_per_step'], d['value'], r.get('kernel'), r.get('kernel_ms'), r.get('frac'), (r.get('reference_algorithm') or {}).get('kernel_ms'))" | tee -a "$OUT/ab.txt"
    done
  done
done
