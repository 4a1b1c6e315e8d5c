#!/bin/bash
OUT=gpurun_out/r3_run5
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sparse_hit_list or bucketed or ordered_hits_only_search_is_exact" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
Q="--no-cpu --no-e2e --general-steps 0 --steps 5 --warmup 2 --no-secondary"
python bench.py $Q > $OUT/enc4.json 2> $OUT/enc4.err; echo "rc $?"; tail -3 $OUT/enc4.err
AWFM_GPU_ENCODE_ONE=1 python bench.py $Q > $OUT/enc1.json 2> $OUT/enc1.err
PROFILE_PASSES="fetch write l2 sq sq2 tcp" bash scripts/profile_bench.sh default 2>&1 | tail -25
python - <<'PY'
import json,glob,os,csv
for f in sorted(glob.glob("gpurun_out/r3_run5/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f),"FAILED",e); continue
    r=d["roofline"]
    print(os.path.basename(f), d["value"], d["ms_per_step"], "search", r.get("call",{}).get("ms"), "dom", r["kernel_ms"], "frac", r["frac"], "digests", d["digests"]["status"], d["config"]["locate_kernels_ms"])
PY
