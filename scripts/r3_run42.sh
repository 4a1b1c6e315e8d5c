#!/bin/bash
OUT=gpurun_out/r3_run42
mkdir -p $OUT
Q="--no-cpu --no-e2e --general-steps 0 --no-secondary --workload planted --steps 4 --warmup 2"
for b in 11 13 15; do
AWFM_GPU_ORDERED_SORT=rocprim AWFM_GPU_ORDER_KEY_BITS=$b python bench.py $Q > $OUT/b$b.json 2> $OUT/b$b.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/b$b.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("rocprim order bits $b planted", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], d["digests"]["status"])
except Exception as ex:
    print("bits $b failed", ex, open("$OUT/b$b.err").read()[-600:])
PY
done
AWFM_GPU_ORDERED_SORT=rocprim python bench.py $Q > $OUT/b15d.json 2> $OUT/b15d.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/b15d.json") if l.startswith("{")][-1]); r=d["roofline"]
print("rocprim default planted", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], d["digests"]["status"])
PY
