#!/bin/bash
# usage: ab_quick.sh name [ENV=..]... -- bench args   (one compact bench line)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
name=$1; shift
envs=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done; shift
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
