#!/bin/bash
OUT=gpurun_out/r3_run26
mkdir -p $OUT
python - <<'PY'
import os
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try: print("cgroup cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print(e)
PY
timeout 900 python -m pytest tests/test_gpu_aos.py tests/test_gpu_budget.py -m gpu -x -q 2>&1 | tail -3
ls tests | head -40
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
grep "awfm aos" $OUT/$name.err | tail -6
}
run chunk2m
run onechunk AWFM_GPU_AOS_CHUNK=100000000
run chunk1m AWFM_GPU_AOS_CHUNK=1000000
run chunk4m AWFM_GPU_AOS_CHUNK=4000000
run lanes3 AWFM_GPU_DEVICES=0,0,0
run lanes3_1m AWFM_GPU_DEVICES=0,0,0 AWFM_GPU_AOS_CHUNK=1000000
