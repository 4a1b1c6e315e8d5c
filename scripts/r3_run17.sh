#!/bin/bash
OUT=gpurun_out/r3_run17
mkdir -p $OUT
Q="--no-cpu --no-e2e --general-steps 0 --no-secondary --mode count"
for p in 0 2 1; do
AWFM_BENCH_DIGESTS=/nonexistent AWFM_GPU_PROBE_PAIR_STEPS=$p python bench.py $Q --steps 10 --warmup 3 > $OUT/probe_$p.json 2> $OUT/probe_$p.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/probe_$p.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("probe $p", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"])
except Exception as e:
    print("probe $p failed", e, open("$OUT/probe_$p.err").read()[-400:])
PY
done
