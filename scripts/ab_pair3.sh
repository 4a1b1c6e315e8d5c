#!/bin/bash
cd "$(dirname "$0")/.."
source scripts/ab_lib.sh
python -m pytest tests/test_gpu_pair.py -x -q 2>&1 | tail -2
run lf2_planted -- --workload planted --no-e2e
run lf2_default -- --no-e2e --no-secondary
run lf2_count -- --no-e2e --mode count
