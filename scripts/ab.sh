#!/bin/bash
# A/B on one box: avxwindowfmindex_amd/libawfmindex_amd_prev.so (A) against the current build (B); args = bench.py args
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
show='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], "Mkmers/s", d["ms_per_step"], "ms/step, search", d["roofline"]["kernel_ms"], "locate", d["config"]["locate_kernels_ms"])'
for rep in 1 2; do
  AWFM_LIB_PATH=$ROOT/avxwindowfmindex_amd/libawfmindex_amd_prev.so python3 bench.py --no-cpu "$@" 2>/dev/null | tail -1 | python3 -c "$show" A
  python3 bench.py --no-cpu "$@" 2>/dev/null | tail -1 | python3 -c "$show" B
done
