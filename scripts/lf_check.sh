#!/bin/bash
OUT=gpurun_out/lf_check
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lookup_first or sparse or search_order or ordered_hits_only_search_is_exact" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest.log
Q="--no-cpu --no-e2e --general-steps 0 --no-secondary"
for m in "" 0; do
for w in random count planted; do
if [ $w = random ]; then A="--steps 10 --warmup 3"; elif [ $w = count ]; then A="--mode count --steps 10 --warmup 3"; else A="--workload planted --steps 4 --warmup 2"; fi
if [ -z "$m" ]; then E=""; else E="AWFM_GPU_LOOKUP_FIRST=$m"; fi
env $E python bench.py $Q $A > $OUT/${w}_lf$m.json 2> $OUT/${w}_lf$m.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/${w}_lf$m.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("$w lookup_first='$m'", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], d["digests"]["status"])
except Exception as ex:
    print("$w lf $m failed", ex, open("$OUT/${w}_lf$m.err").read()[-1200:])
PY
done
done
