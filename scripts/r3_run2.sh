#!/bin/bash
# round-3 GPU call 2: whole GPU suite with the new defaults, deep-table build time, genome-shaped text at full size,
# planted key-width sweep, digests of the default batches
OUT=gpurun_out/r3_run2
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
AWFM_VERBOSE=1 python bench.py --steps 5 --warmup 2 --record-digests $OUT/digests.json > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default rc $?"; grep -E "deep seed|doubling|awfm build" $OUT/bench_default.err | tail -12
Q="--no-cpu --no-e2e --no-secondary --general-steps 0"
for r in 1 2 3 4 5 6 7; do
  python bench.py $Q --steps 1 --warmup 1 --query-offset ${r}e8 --record-digests $OUT/digests.json > $OUT/shard_$r.json 2> $OUT/shard_$r.err
done
python bench.py $Q --steps 2 --warmup 1 --workload mixed --record-digests $OUT/digests.json > $OUT/mixed.json 2> $OUT/mixed.err
python bench.py $Q --steps 2 --warmup 1 --mode count --record-digests $OUT/digests.json > $OUT/count.json 2> $OUT/count.err
for bits in 15 12 11; do
  AWFM_GPU_ORDERED_WIDE=1 AWFM_GPU_ORDER_KEY_BITS=$bits python bench.py $Q --workload planted --steps 3 --warmup 1 > $OUT/planted_keybits_$bits.json 2> $OUT/planted_keybits_$bits.err
done
AWFM_VERBOSE=1 python bench.py $Q --text repetitive --steps 3 --warmup 1 > $OUT/rep_random.json 2> $OUT/rep_random.err
echo "rep random rc $?"; grep -E "doubling|awfm build|deep seed" $OUT/rep_random.err | tail -40
python bench.py $Q --text repetitive --workload planted --steps 3 --warmup 1 > $OUT/rep_planted.json 2> $OUT/rep_planted.err
echo "rep planted rc $?"; tail -3 $OUT/rep_planted.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r3_run2/*.json")):
    if f.endswith("digests.json"): continue
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f),"FAILED",e); continue
    r=d["roofline"]
    print(os.path.basename(f), d["value"], d["ms_per_step"], "search", r.get("call",{}).get("ms"), "dom", r["kernel_ms"], "frac", r["frac"], "digests", d["digests"]["status"], "build", d["config"]["index_build_s"], "deep", d["config"]["device_seed_k"])
PY
