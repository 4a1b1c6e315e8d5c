# source me: run <name> [ENV=..]... -- <bench args>  prints one compact bench line
mkdir -p gpurun_out
run() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --steps 10 --warmup 3 --no-cpu "$@" > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/ab_$name.json") if l.startswith("{")][-1])
    r=d["roofline"]; e=d.get("end_to_end") or {}
    print("%-22s %9.1f Mk/s  step %7.3f ms  search %7.3f ms  dom %s  locate %7.3f ms  frac %.3f  build %.1fs  e2e %s" % ("$name", d["value"], d["ms_per_step"], r["kernel_ms"], (r.get("dominant_kernel") or {}).get("ms"), d["config"]["locate_kernels_ms"], r["frac"], d["config"]["index_build_s"], {k:(v["value"], v["ms"]) for k,v in e.items() if isinstance(v, dict)}))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/ab_$name.err").read()[-1500:])
PY
}
