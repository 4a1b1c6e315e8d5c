#!/bin/bash
OUT=gpurun_out/r3_run23
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_stream.py tests/test_gpu_budget.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.log
PROFILE_PASSES="fetch l2 rdreq" bash scripts/profile_bench.sh r3x --steps 5 --warmup 2 --general-steps 0 2>&1 | tail -30
python - <<'PY'
import json
print(open("gpurun_out/prof_r3x/bench_trace.log").read()[-300:])
PY
