#!/bin/bash
OUT=gpurun_out/r3_run27
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.log
Q="--no-cpu --general-steps 0 --no-secondary --steps 3 --warmup 1"
run() { # name env...
name=$1; shift
env "$@" AWFM_GPU_AOS_TRACE=1 python bench.py $Q > $OUT/$name.json 2> $OUT/$name.err
python - $OUT/$name.json "$name" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], d["end_to_end"]["aos_drop_in"])
except Exception as e:
    print(sys.argv[2], "failed", e, open(sys.argv[1][:-5]+".err").read()[-800:])
PY
}
run default
run chunk512k AWFM_GPU_AOS_CHUNK=524288
run lanes4 AWFM_GPU_DEVICES=0,0,0,0
run lanes4_512k AWFM_GPU_DEVICES=0,0,0,0 AWFM_GPU_AOS_CHUNK=524288
run count_default -- 
python bench.py --mode count $Q > $OUT/count.json 2> $OUT/count.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3_run27/count.json") if l.startswith("{")][-1]); print("count", d["end_to_end"]["aos_drop_in"], d["value"])
PY
