#!/bin/bash
# round-3 GPU call 3: the hand-written count + partition front end: parity tests, then A/B against the rocPRIM sort
OUT=gpurun_out/r3_run3
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_pair.py tests/test_gpu_stream.py tests/test_gpu_fuzz.py -m gpu -x -q -k "ordered or bucketed or packed or stream or fuzz or pair" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
Q="--no-cpu --no-e2e --no-secondary --general-steps 0 --steps 5 --warmup 2"
python bench.py $Q > $OUT/partition_locate.json 2> $OUT/partition_locate.err; echo "rc $?"; tail -3 $OUT/partition_locate.err
AWFM_GPU_ORDERED_SORT=rocprim python bench.py $Q > $OUT/rocprim_locate.json 2> $OUT/rocprim_locate.err
python bench.py $Q --workload planted --steps 3 > $OUT/partition_planted.json 2> $OUT/partition_planted.err
AWFM_GPU_ORDERED_SORT=rocprim python bench.py $Q --workload planted --steps 3 > $OUT/rocprim_planted.json 2> $OUT/rocprim_planted.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $Q > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do head -16 $f | cut -c1-220; done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r3_run3/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f),"FAILED",e); continue
    r=d["roofline"]
    print(os.path.basename(f), d["value"], d["ms_per_step"], "search", r.get("call",{}).get("ms"), "dom", r["kernel_ms"], "frac", r["frac"], "digests", d["digests"]["status"])
PY
