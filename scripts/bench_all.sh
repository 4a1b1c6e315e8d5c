#!/bin/bash
# every bench.py configuration quoted in DESIGN.md, one JSON line each -> gpurun_out/bench_<tag>/bench_<name>.json
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/bench_$TAG
mkdir -p "$OUT"
cd "$ROOT"
run() { # name [ENV=..]... -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python3 bench.py "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/bench_$name.json"
}
run default --
run count -- --mode count
run planted -- --workload planted
run mixed -- --workload mixed
run amino -- --alphabet amino
run general AWFM_GPU_ORDERED=0 -- --mode count --no-cpu --no-e2e
run general_letters AWFM_GPU_ORDERED=0 AWFM_GPU_GENERAL_NO_PAIR=1 -- --mode count --no-cpu --no-e2e
run nopair_default AWFM_GPU_PAIR=0 -- --no-cpu --no-e2e
run nopair_planted AWFM_GPU_PAIR=0 -- --workload planted --no-cpu --no-e2e
run deep14 -- --device-seed-k 14 --no-cpu --no-e2e
run planted_dense -- --workload planted --device-dense-sa --no-cpu --no-e2e
python3 - "$OUT" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
        continue
    r, c = d["roofline"], d.get("cpu_baseline") or {}
    print(f"{os.path.basename(f)[6:-5]:22s} {d['value']:9.1f} Mkmers/s  {d['ms_per_step']:7.2f} ms/step  search {r['kernel_ms']:6.2f} ms  "
          f"frac {r['frac']:.3f}  locate {d['config'].get('locate_kernels_ms', 0):6.2f} ms  cpu {c.get('value')} on {c.get('cores')}")
PY
