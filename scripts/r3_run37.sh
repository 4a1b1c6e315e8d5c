#!/bin/bash
OUT=gpurun_out/r3_run37
mkdir -p $OUT
timeout 2700 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.log
AWFM_VERBOSE=1 python bench.py --no-cpu --general-steps 0 --steps 10 --warmup 3 > $OUT/default.json 2> $OUT/default.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/default.json") if l.startswith("{")][-1]); r=d["roofline"]
print("default", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], "frac", r["frac"], d["digests"]["status"], d["config"]["index_build_s"], d["config"]["device_image_bytes"])
print({k:(v.get("value") if isinstance(v,dict) else v) for k,v in d["end_to_end"].items()})
print(d.get("secondary"))
PY
grep -i "deep seed" $OUT/default.err | tail -5
