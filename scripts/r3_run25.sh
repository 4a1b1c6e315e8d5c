#!/bin/bash
OUT=gpurun_out/r3_run25
mkdir -p $OUT
nproc; free -g | head -2
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
grep "awfm aos" $OUT/$name.err | tail -12
}
run chunk2m
run onechunk AWFM_GPU_AOS_CHUNK=100000000
run chunk1m AWFM_GPU_AOS_CHUNK=1000000
run lanes3 AWFM_GPU_DEVICES=0,0,0
run lanes3_1m AWFM_GPU_DEVICES=0,0,0 AWFM_GPU_AOS_CHUNK=1000000
