#!/bin/bash
OUT=gpurun_out/r3_run28
mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_bench_multirank.py tests/test_multirank.py tests/test_gpu_pair.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.log
Q="--no-cpu --general-steps 0 --no-secondary --steps 3 --warmup 1"
run() { # name env...
name=$1; shift
env "$@" python bench.py $Q > $OUT/$name.json 2> $OUT/$name.err
python - $OUT/$name.json "$name" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    a=d["end_to_end"]["aos_drop_in"]; print(sys.argv[2], a["value"], a["ms_all"])
except Exception as e:
    print(sys.argv[2], "failed", e, open(sys.argv[1][:-5]+".err").read()[-800:])
PY
}
run default
run lanes2 AWFM_GPU_DEVICES=0,0
run lanes2_2m AWFM_GPU_DEVICES=0,0 AWFM_GPU_AOS_CHUNK=2097152
run lanes3_2m AWFM_GPU_AOS_CHUNK=2097152
run lanes3_512k AWFM_GPU_AOS_CHUNK=524288
run lanes4 AWFM_GPU_DEVICES=0,0,0,0
run lanes1 AWFM_GPU_DEVICES=0
