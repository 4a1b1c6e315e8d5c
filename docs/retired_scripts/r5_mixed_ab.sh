#!/bin/bash
# A/B on one box: the mixed-length step with whole-line counts / prediction on and off; the genome-shaped text's random batch
# (its shards: the list tail expands hit lists of 10^5)
out=gpurun_out/r5_mixed_ab
mkdir -p $out
run() { # name, env..., -- args
  name=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" python bench.py --no-cpu --no-e2e --no-secondary --general-steps 0 "$@" > $out/$name.json 2> $out/$name.err
  python - "$out/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    sp=(d.get("scaling_proxy") or {}).get("shards",{})
    e8={k:(v.get("8",{}).get("ms_max"), v.get("8",{}).get("efficiency")) for k,v in sp.items()}
    print(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"].get("kernel_ms"), d["config"].get("lookup_front"), e8)
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
run mixed_$rep -- --workload mixed
run mixed_nowhole_$rep AWFM_GPU_MIXED_WHOLE_COUNTS=0 -- --workload mixed
run mixed_nopredict_$rep AWFM_GPU_LOOKUP_PREDICT=0 -- --workload mixed
done
run rep_random -- --text repetitive
