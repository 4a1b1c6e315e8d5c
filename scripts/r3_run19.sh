#!/bin/bash
OUT=gpurun_out/r3_run19
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
for b in 6 5 4; do
AWFM_GPU_BLOCKS_PER_CU=$b python bench.py $Q --steps 10 --warmup 3 > $OUT/occ_$b.json 2> $OUT/occ_$b.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/occ_$b.json") if l.startswith("{")][-1]); r=d["roofline"]
print("blocks/CU $b", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], d["digests"]["status"])
PY
done
