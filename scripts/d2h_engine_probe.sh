#!/bin/bash
# which engine downloads run on, and what the packed pipeline makes of it: the pipeline alone (scripts/stream_probe.py:
# GRCh38-sized index, 10^8 planted 21-mers, locate) under the runtime's copy-engine knobs
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for e in "X=0" "GPU_FORCE_BLIT_COPY_SIZE=0" "GPU_FORCE_BLIT_COPY_SIZE=1" "HSA_FORCE_SDMA_SIZE=0" "GPU_BLIT_ENGINE_TYPE=2" "HSA_ENABLE_SDMA_COPY_SIZE_OVERRIDE=0"; do
  echo "== $e"
  env "$e" timeout 300 python3 scripts/stream_probe.py 3.1e9 1e8 planted 2>&1 | grep "^run" | tail -2
done
