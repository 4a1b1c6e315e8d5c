#!/bin/bash
OUT=gpurun_out/r3_run30
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deep_seed or ordered or device_images" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -8 $OUT/pytest.log
Q="--no-cpu --no-e2e --general-steps 0 --no-secondary"
for m in count locate; do
AWFM_VERBOSE=1 python bench.py $Q --mode $m --steps 10 --warmup 3 > $OUT/$m.json 2> $OUT/$m.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/$m.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("$m", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], "frac", r["frac"], r["compulsory_bytes"], d["digests"]["status"], "build", d["config"]["device_seed_build_s"])
except Exception as e:
    print("$m failed", e, open("$OUT/$m.err").read()[-1500:])
PY
done
python bench.py $Q --workload planted --steps 5 --warmup 2 > $OUT/planted.json 2> $OUT/planted.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/planted.json") if l.startswith("{")][-1]); r=d["roofline"]
print("planted", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], d["digests"]["status"])
PY
