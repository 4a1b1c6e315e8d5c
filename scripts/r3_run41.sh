#!/bin/bash
OUT=gpurun_out/r3_run41
mkdir -p $OUT
Q="--no-cpu --no-e2e --general-steps 0 --no-secondary"
for v in "" _nt; do
for w in planted random; do
if [ $w = random ]; then A="--mode count --steps 10 --warmup 3"; else A="--workload planted --steps 4 --warmup 2"; fi
AWFM_LIB_PATH=$PWD/avxwindowfmindex_amd/libawfmindex_amd$v.so python bench.py $Q $A > $OUT/v${v}_$w.json 2> $OUT/v${v}_$w.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/v${v}_$w.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("variant '$v' $w", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], d["digests"]["status"])
except Exception as ex:
    print("variant $v failed", ex, open("$OUT/v${v}_$w.err").read()[-600:])
PY
done
done
