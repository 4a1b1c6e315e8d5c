#!/bin/bash
cd "$(dirname "$0")/.."
AWFM_GPU_STREAM_DIRECT=1 python -m pytest tests/test_gpu_stream.py -x -q 2>&1 | tail -2
echo "== copy"; python scripts/stream_probe.py 3.1e9 1e8 planted 2>&1 | grep "^run [123]"
for b in 1 2 8; do echo "== direct, finish blocks/CU $b"; AWFM_GPU_STREAM_DIRECT=1 AWFM_GPU_FINISH_BLOCKS=$b python scripts/stream_probe.py 3.1e9 1e8 planted 2>&1 | grep "^run [123]"; done
echo "== direct split mode"; AWFM_GPU_STREAM_MODE=split AWFM_GPU_STREAM_DIRECT=1 python scripts/stream_probe.py 3.1e9 1e8 planted 2>&1 | grep "^run [123]"
