#!/bin/bash
# pair image on/off, and the knobs around it: compact bench lines of the main workloads on one box
cd "$(dirname "$0")/.."
source scripts/ab_lib.sh
run pair_default -- --no-e2e --no-secondary
run nopair_default AWFM_GPU_PAIR=0 -- --no-e2e --no-secondary
run pair_planted -- --workload planted --no-e2e
run nopair_planted AWFM_GPU_PAIR=0 -- --workload planted --no-e2e
run pair_super_global AWFM_GPU_PAIR_SUPER=global -- --no-e2e --no-secondary
for b in 3 4 5 6; do run pair_occ${b}_count AWFM_GPU_BLOCKS_PER_CU=$b -- --mode count --no-e2e; done
