#!/bin/bash
OUT=gpurun_out/r3_run33
mkdir -p $OUT
Q="--no-cpu --no-e2e --general-steps 0 --no-secondary --mode count"
for v in "" _w7 _w8; do
for c in 4 2 8; do
AWFM_GPU_CHUNKS_PER_TICKET=$c AWFM_LIB_PATH=$PWD/avxwindowfmindex_amd/libawfmindex_amd$v.so python bench.py $Q --steps 10 --warmup 3 > $OUT/v${v}_$c.json 2> $OUT/v${v}_$c.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/v${v}_$c.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("variant '$v' chunks/ticket $c", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], d["digests"]["status"])
except Exception as ex:
    print("variant $v failed", ex, open("$OUT/v${v}_$c.err").read()[-600:])
PY
done
done
