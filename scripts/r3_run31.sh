#!/bin/bash
OUT=gpurun_out/r3_run31
mkdir -p $OUT
Q="--no-cpu --no-e2e --general-steps 0 --no-secondary --mode count"
for v in "" _NOBIG _ATLOAD; do
for e in 1 0; do
AWFM_GPU_DEEP_NEXT=$e AWFM_LIB_PATH=$PWD/avxwindowfmindex_amd/libawfmindex_amd$v.so python bench.py $Q --steps 10 --warmup 3 > $OUT/v${v}_$e.json 2> $OUT/v${v}_$e.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/v${v}_$e.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("variant '$v' next $e", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], r["compulsory_bytes"], d["digests"]["status"])
except Exception as ex:
    print("variant $v failed", ex, open("$OUT/v${v}_$e.err").read()[-600:])
PY
done
done
