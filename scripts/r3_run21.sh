#!/bin/bash
OUT=gpurun_out/r3_run21
mkdir -p $OUT
Q="--no-cpu --no-e2e --general-steps 0 --no-secondary --mode count"
for v in "" _w7a1; do
lib=$PWD/avxwindowfmindex_amd/libawfmindex_amd$v.so
for k in 13 15 16; do
AWFM_LIB_PATH=$lib timeout 300 python bench.py $Q --device-seed-k $k --steps 10 --warmup 3 > $OUT/v${v}_$k.json 2> $OUT/v${v}_$k.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/v${v}_$k.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("variant '$v' depth $k", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], d["digests"]["status"], r["compulsory_bytes"], d["config"]["device_seed_build_s"])
except Exception as e:
    print("variant $v depth $k failed", e, open("$OUT/v${v}_$k.err").read()[-600:])
PY
done
done
