#!/bin/bash
# smoke, the whole GPU suite, every bench line and the rocprofv3 passes behind profiles/<round>: gpurun -- bash scripts/full_gpu_check.sh
OUT=gpurun_out/full_gpu_check
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
AWFM_COMMIT=${AWFM_COMMIT:-$(cat .git_head 2>/dev/null || echo unknown)} bash scripts/refresh_profiles.sh ${1:-r6}
