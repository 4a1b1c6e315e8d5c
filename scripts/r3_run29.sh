#!/bin/bash
OUT=gpurun_out/r3_run29
mkdir -p $OUT
Q="--no-cpu --no-e2e --general-steps 0 --no-secondary --mode count"
for v in "" _probe; do
AWFM_BENCH_DIGESTS=/nonexistent AWFM_LIB_PATH=$PWD/avxwindowfmindex_amd/libawfmindex_amd$v.so python bench.py $Q --steps 10 --warmup 3 > $OUT/v$v.json 2> $OUT/v$v.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/v$v.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("variant '$v'", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], r["compulsory_bytes"])
except Exception as e:
    print("variant $v failed", e, open("$OUT/v$v.err").read()[-600:])
PY
done
