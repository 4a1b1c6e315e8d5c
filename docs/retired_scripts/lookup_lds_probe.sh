#!/bin/bash
# where the lookup kernels keep the pair image's superblock bases, and how many workgroups per CU -> gpurun_out/r4/lookup_lds_probe.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r4
OUT=gpurun_out/r4/lookup_lds_probe.txt; : > $OUT
run() { # label env... -- args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python3 bench.py "$@" --no-cpu --no-e2e --no-secondary --general-steps 0 --no-shard-proxy --no-dense-form 2>/dev/null | tail -1 > /tmp/lp.json
  python3 - "$label" >> $OUT <<'PY'
import json, sys
d = json.load(open("/tmp/lp.json"))
print(sys.argv[1], d["value"], "Mkmers/s", d["ms_per_step"], "ms", d["roofline"].get("kernel"), d["roofline"].get("kernel_ms"), d["config"].get("dominant_kernel_ms_first_min_max"))
PY
}
run "random lds x7" AWFM_GPU_LOOKUP_PAIR_SUPER=lds --
run "random lds x4" AWFM_GPU_LOOKUP_PAIR_SUPER=lds AWFM_GPU_LOOKUP_BLOCKS_PER_CU=4 --
run "random global x7" AWFM_GPU_LOOKUP_PAIR_SUPER=global --
run "random global x8" AWFM_GPU_LOOKUP_PAIR_SUPER=global AWFM_GPU_LOOKUP_BLOCKS_PER_CU=8 --
run "mixed lds" AWFM_GPU_MIXED_LOOKUP=1 AWFM_GPU_LOOKUP_PAIR_SUPER=lds -- --workload mixed
run "mixed global" AWFM_GPU_MIXED_LOOKUP=1 AWFM_GPU_LOOKUP_PAIR_SUPER=global -- --workload mixed
cat $OUT
