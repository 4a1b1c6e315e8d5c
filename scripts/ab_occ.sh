#!/bin/bash
# EXPERIMENT: is the ordered search latency-bound?  resident workgroups per CU 2/4/6/8 (x4 waves)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for b64 in 0 1; do
for occ in 2 4 6 8; do
  envs="AWFM_GPU_BLOCKS_PER_CU=$occ"
  [ $b64 = 1 ] && envs="$envs AWFM_GPU_B64=1"
  env $envs python bench.py --steps 10 --warmup 3 --no-cpu --mode count > gpurun_out/occ.json 2> gpurun_out/occ.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/occ.json") if l.startswith("{")][-1])
    print("b64=$b64 blocks/CU=$occ", d["value"], "search", d["roofline"]["kernel_ms"], d["roofline"].get("dominant_kernel"))
except Exception as e:
    print("FAILED", e); print(open("gpurun_out/occ.err").read()[-1500:])
PY
done
done
