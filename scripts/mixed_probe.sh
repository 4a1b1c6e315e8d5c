#!/bin/bash
# mixed-length batches: the lookup-first kernel forced on / off, two length ranges -> gpurun_out/r4/mixed_probe.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r4
OUT=gpurun_out/r4/mixed_probe.txt; : > $OUT
for lens in "8 30" "18 30" "8 15"; do
  for mode in 1 0; do
    AWFM_GPU_MIXED_LOOKUP=$mode python3 bench.py --workload mixed --mixed-lengths $lens --no-cpu --no-e2e --general-steps 0 --no-shard-proxy 2>/dev/null | tail -1 > /tmp/mp.json
    python3 - "$lens" $mode >> $OUT <<'PY'
import json, sys
d = json.load(open("/tmp/mp.json"))
print(sys.argv[1], "lookup", sys.argv[2], d["value"], "Mkmers/s", d["ms_per_step"], "ms", d["roofline"].get("kernel"), d["roofline"].get("kernel_ms"), (d.get("parity") or {}).get("status"))
PY
  done
done
cat $OUT
