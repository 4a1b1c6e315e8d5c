#!/bin/bash
OUT=gpurun_out/r3_run11
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
bash scripts/refresh_profiles.sh r3
