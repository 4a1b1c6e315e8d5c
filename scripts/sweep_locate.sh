#!/bin/bash
# planted workload (>= 1 hit per k-mer): walk-kernel variants and residency, one box
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
show='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], "Mkmers/s", d["ms_per_step"], "ms/step, search", d["roofline"]["kernel_ms"], "locate", d["config"]["locate_kernels_ms"])'
for k in g4 g8 g2 g1 g4; do
  AWFM_GPU_LOCATE_KERNEL=$k python3 bench.py --no-cpu --workload planted --steps 3 --warmup 1 2>/dev/null | tail -1 | python3 -c "$show" "locate=$k"
done
