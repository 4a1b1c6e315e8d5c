#!/bin/bash
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_stream.py -x -q 2>&1 | tail -2
for lag in 2 3; do for m in split slots; do
 echo "== lag $lag mode $m planted"; AWFM_GPU_STREAM_LAG=$lag AWFM_GPU_STREAM_MODE=$m python scripts/stream_probe.py 3.1e9 1e8 planted 2>&1 | grep "^run [123]"
done; done
