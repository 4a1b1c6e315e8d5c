#!/bin/bash
# round-3 GPU call 1: new tests, the default line with the new roofline record, key-width and table-depth sweeps
OUT=gpurun_out/r3_run1
mkdir -p $OUT
python -m pytest tests/test_bench_multirank.py tests/test_gpu_stream.py -m gpu -x -q > $OUT/pytest_new.log 2>&1
echo "pytest new rc $?" >> $OUT/pytest_new.log
tail -3 $OUT/pytest_new.log
python bench.py --steps 5 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default rc $?"
Q="--mode count --no-cpu --no-e2e --no-secondary --general-steps 0 --steps 5 --warmup 2"
for bits in 15 14 13 12 11 10; do
  AWFM_GPU_ORDERED_WIDE=1 AWFM_GPU_ORDER_KEY_BITS=$bits python bench.py $Q > $OUT/keybits_$bits.json 2> $OUT/keybits_$bits.err
done
for k in 13 14 15; do
  python bench.py $Q --device-seed-k $k > $OUT/deep_$k.json 2> $OUT/deep_$k.err
  python bench.py --workload planted --no-cpu --no-e2e --general-steps 0 --steps 3 --warmup 1 --device-seed-k $k > $OUT/deep_planted_$k.json 2> $OUT/deep_planted_$k.err
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r3_run1/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f),"FAILED",e); continue
    r=d["roofline"]
    print(os.path.basename(f), d["value"], d["ms_per_step"], "search", r.get("call",{}).get("ms"), "dom", r["kernel_ms"], "frac", r["frac"], "digests", d["digests"]["status"])
PY
