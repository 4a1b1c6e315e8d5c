#!/bin/bash
OUT=gpurun_out/r3_run40
mkdir -p $OUT
Q="--no-cpu --no-e2e --general-steps 0 --no-secondary --workload mixed --steps 5 --warmup 2"
for l in "8 30" "12 30" "16 30" "18 30" "8 15"; do
set -- $l
python bench.py $Q --mixed-lengths $1 $2 > $OUT/m_$1_$2.json 2> $OUT/m_$1_$2.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/m_$1_$2.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("mixed $1..$2", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], r["frac"], d["digests"]["status"], r["compulsory"])
except Exception as ex:
    print("mixed $1 $2 failed", ex, open("$OUT/m_$1_$2.err").read()[-800:])
PY
done
