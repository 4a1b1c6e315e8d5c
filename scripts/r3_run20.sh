#!/bin/bash
OUT=gpurun_out/r3_run20
mkdir -p $OUT
Q="--no-cpu --no-e2e --general-steps 0 --no-secondary --mode count"
for v in "" _w7a0 _w7a1 _w8a0 _w8a1; do
lib=$PWD/avxwindowfmindex_amd/libawfmindex_amd$v.so
for c in 4 8; do
AWFM_LIB_PATH=$lib AWFM_GPU_CHUNKS_PER_TICKET=$c python bench.py $Q --steps 10 --warmup 3 > $OUT/v${v}_$c.json 2> $OUT/v${v}_$c.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/v${v}_$c.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("variant '$v' chunks $c", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], d["digests"]["status"])
except Exception as e:
    print("variant $v failed", e, open("$OUT/v${v}_$c.err").read()[-400:])
PY
done
done
