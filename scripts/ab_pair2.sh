#!/bin/bash
cd "$(dirname "$0")/.."
source scripts/ab_lib.sh
run lds_default AWFM_GPU_PAIR_SUPER=lds -- --no-e2e
run glb_default AWFM_GPU_PAIR_SUPER=global -- --no-e2e
run lds_planted AWFM_GPU_PAIR_SUPER=lds -- --workload planted --no-e2e
run glb_planted AWFM_GPU_PAIR_SUPER=global -- --workload planted --no-e2e
