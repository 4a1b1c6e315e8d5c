#!/bin/bash
# bench lines of the main workloads and lane variants, compact
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --steps 10 --warmup 3 --no-cpu "$@" > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/ab_$name.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("%-28s %9.1f Mk/s  step %7.3f ms  search %7.3f ms  dom %s  locate %7.3f ms  frac %.3f" % ("$name", d["value"], d["ms_per_step"], r["kernel_ms"], (r.get("dominant_kernel") or {}).get("ms"), d["config"]["locate_kernels_ms"], r["frac"]))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/ab_$name.err").read()[-800:])
PY
}
run default -- 
run default_ordG2 AWFM_GPU_ORDERED_LANES=2 --
run planted -- --workload planted
run planted_locG2 AWFM_GPU_LOCATE_KERNEL=g2 -- --workload planted
run planted_locG1 AWFM_GPU_LOCATE_KERNEL=g1 -- --workload planted
run general_g4 AWFM_GPU_ORDERED=0 AWFM_GPU_KERNEL=g4 -- --mode count
run general_g2 AWFM_GPU_ORDERED=0 AWFM_GPU_KERNEL=g2 -- --mode count
run general_g1 AWFM_GPU_ORDERED=0 AWFM_GPU_KERNEL=g1 -- --mode count
run mixed -- --workload mixed
run amino_g4 -- --alphabet amino
run amino_g2 AWFM_GPU_KERNEL=g2 -- --alphabet amino
run amino_planted -- --alphabet amino --workload planted
