#!/bin/bash
OUT=gpurun_out/r3_run32
mkdir -p $OUT
Q="--no-cpu --no-e2e --general-steps 0 --no-secondary --mode count"
for k in 14 15 16; do
AWFM_LIB_PATH=$PWD/avxwindowfmindex_amd/libawfmindex_amd_ATLOAD.so python bench.py $Q --device-seed-k $k --steps 10 --warmup 3 > $OUT/k$k.json 2> $OUT/k$k.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/k$k.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("depth $k", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], r["compulsory_bytes"], r["compulsory"]["deep_table_lines"], r["compulsory"]["pair_level_lines"], d["digests"]["status"])
except Exception as ex:
    print("depth $k failed", ex, open("$OUT/k$k.err").read()[-600:])
PY
done
