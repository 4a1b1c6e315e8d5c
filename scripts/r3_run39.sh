#!/bin/bash
OUT=gpurun_out/r3_run39
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "ordered or deep_seed or fuzz or search_order or sparse or bucket" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $OUT/pytest.log
Q="--no-cpu --no-e2e --general-steps 0 --no-secondary"
for kern in units chunks; do
for w in random planted mixed; do
if [ $w = random ]; then A="--mode count --steps 10 --warmup 3"; elif [ $w = planted ]; then A="--workload planted --steps 4 --warmup 2"; else A="--workload mixed --steps 5 --warmup 2"; fi
AWFM_GPU_BUCKET_KERNEL=$kern python bench.py $Q $A > $OUT/${kern}_$w.json 2> $OUT/${kern}_$w.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/${kern}_$w.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("$kern $w", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], d["digests"]["status"])
except Exception as ex:
    print("$kern $w failed", ex, open("$OUT/${kern}_$w.err").read()[-800:])
PY
done
done
