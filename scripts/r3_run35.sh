#!/bin/bash
OUT=gpurun_out/r3_run35
mkdir -p $OUT
Q="--no-cpu --no-e2e --general-steps 0 --no-secondary"
for v in _w6g1 "" _w6g8 _w7g4 _w7g8; do
for c in 1 2 4 8; do
for w in random planted; do
if [ $w = random ]; then A="--mode count --steps 10 --warmup 3"; else A="--workload planted --steps 4 --warmup 2"; fi
AWFM_GPU_CHUNKS_PER_TICKET=$c AWFM_LIB_PATH=$PWD/avxwindowfmindex_amd/libawfmindex_amd$v.so python bench.py $Q $A > $OUT/v${v}_${c}_$w.json 2> $OUT/v${v}_${c}_$w.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/v${v}_${c}_$w.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("variant '$v' chunks/ticket $c $w", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], d["digests"]["status"])
except Exception as ex:
    print("variant $v failed", ex, open("$OUT/v${v}_${c}_$w.err").read()[-600:])
PY
done
done
done
