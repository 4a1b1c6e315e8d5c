#!/bin/bash
OUT=gpurun_out/r3_run24
mkdir -p $OUT
Q="--no-cpu --no-e2e --general-steps 0 --no-secondary --mode count"
for a in default uncached finegrained; do
for v in "" _l1 _l2 _l3; do
lib=$PWD/avxwindowfmindex_amd/libawfmindex_amd$v.so
AWFM_GPU_DEEP_ALLOC=$a AWFM_LIB_PATH=$lib timeout 300 python bench.py $Q --steps 10 --warmup 3 > $OUT/v${v}_$a.json 2> $OUT/v${v}_$a.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/v${v}_$a.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("alloc $a load '$v'", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], d["digests"]["status"], d["config"]["device_seed_k"], d["config"]["device_seed_build_s"])
except Exception as e:
    print("alloc $a load '$v' failed", e, open("$OUT/v${v}_$a.err").read()[-600:])
PY
done
done
