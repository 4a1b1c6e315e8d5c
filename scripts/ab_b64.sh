#!/bin/bash
# EXPERIMENT: ordered search on 64-byte blocks vs 128-byte blocks, lanes per query 4/2/1
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for cfg in "base:" "b64g4:AWFM_GPU_B64=1 AWFM_GPU_ORDERED_LANES=4" "b64g2:AWFM_GPU_B64=1 AWFM_GPU_ORDERED_LANES=2" "b64g1:AWFM_GPU_B64=1 AWFM_GPU_ORDERED_LANES=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for wl in random planted; do
    env $envs python bench.py --steps 10 --warmup 3 --cpu-seconds 2 --workload $wl > gpurun_out/ab_${name}_${wl}.json 2> gpurun_out/ab_${name}_${wl}.err
    python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/ab_${name}_${wl}.json") if l.startswith("{")][-1])
    print("${name} ${wl}", d["value"], "ms/step", d["ms_per_step"], "search", d["roofline"]["kernel_ms"], d["roofline"].get("dominant_kernel"), "locate", d["config"]["locate_kernels_ms"])
except Exception as e:
    print("${name} ${wl} FAILED", e); print(open("gpurun_out/ab_${name}_${wl}.err").read()[-1500:])
PY
  done
done
