#!/bin/bash
OUT=gpurun_out/r3_run18
mkdir -p $OUT
Q="--no-cpu --no-e2e --general-steps 0 --no-secondary"
for c in 1 2 4 8 16; do
AWFM_GPU_CHUNKS_PER_TICKET=$c python bench.py $Q --mode count --steps 10 --warmup 3 > $OUT/count_$c.json 2> $OUT/count_$c.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/count_$c.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("chunks/ticket $c count", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], d["digests"]["status"])
except Exception as e:
    print("$c failed", e, open("$OUT/count_$c.err").read()[-400:])
PY
done
for c in 1 4 8; do
AWFM_GPU_CHUNKS_PER_TICKET=$c python bench.py $Q --workload planted --steps 3 > $OUT/planted_$c.json 2> $OUT/planted_$c.err
AWFM_GPU_CHUNKS_PER_TICKET=$c python bench.py $Q --workload mixed --steps 3 > $OUT/mixed_$c.json 2> $OUT/mixed_$c.err
python - <<PY
import json
for n in ("planted","mixed"):
  try:
    d=json.loads([l for l in open("$OUT/%s_$c.json" % n) if l.startswith("{")][-1]); r=d["roofline"]
    print("chunks/ticket $c", n, d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], d["digests"]["status"])
  except Exception as e:
    print("$c", n, "failed", e)
PY
done
