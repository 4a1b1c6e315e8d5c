#!/bin/bash
OUT=gpurun_out/r3_run6
mkdir -p $OUT
python -m pytest tests/test_gpu_stream.py tests/test_gpu_budget.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
