#!/bin/bash
OUT=gpurun_out/r3_run14
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_pair.py -m gpu -x -q -k "mixed or ordered or fuzz or search_order or pair" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
Q="--no-cpu --no-e2e --general-steps 0 --no-secondary"
for v in 7 6 7 6; do
  lib=avxwindowfmindex_amd/libawfmindex_amd.so; [ $v = 6 ] && lib=avxwindowfmindex_amd/libawfmindex_amd_w6.so
  AWFM_LIB_PATH=$PWD/$lib python bench.py $Q --steps 10 --warmup 3 > $OUT/locate_w$v.json 2> $OUT/locate_w$v.err
  python - <<PY
import json
d=json.loads([l for l in open("$OUT/locate_w$v.json") if l.startswith("{")][-1]); r=d["roofline"]
print("waves $v locate", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"])
PY
done
for v in 7 6; do
  lib=avxwindowfmindex_amd/libawfmindex_amd.so; [ $v = 6 ] && lib=avxwindowfmindex_amd/libawfmindex_amd_w6.so
  AWFM_LIB_PATH=$PWD/$lib python bench.py $Q --workload mixed --steps 3 > $OUT/mixed_w$v.json 2> $OUT/mixed_w$v.err
  AWFM_LIB_PATH=$PWD/$lib python bench.py $Q --workload planted --steps 3 > $OUT/planted_w$v.json 2> $OUT/planted_w$v.err
  python - <<PY
import json
for n in ("mixed","planted"):
    d=json.loads([l for l in open("$OUT/%s_w$v.json" % n) if l.startswith("{")][-1]); r=d["roofline"]
    print("waves $v", n, d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], d["digests"]["status"])
PY
done
AWFM_GPU_ORDERED_SORT=rocprim python bench.py $Q --workload mixed --steps 3 > $OUT/mixed_rocprim.json 2> $OUT/mixed_rocprim.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3_run14/mixed_rocprim.json") if l.startswith("{")][-1]); r=d["roofline"]
print("mixed rocprim", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], d["digests"]["status"])
PY
