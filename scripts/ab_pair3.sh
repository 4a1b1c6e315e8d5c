#!/bin/bash
cd "$(dirname "$0")/.."
source scripts/ab_lib.sh
for i in 1 2; do
run inter${i}_planted AWFM_LIB_PATH=$PWD/avxwindowfmindex_amd/libawfm_inter.so -- --workload planted --no-e2e
run split${i}_planted -- --workload planted --no-e2e
run inter${i}_count AWFM_LIB_PATH=$PWD/avxwindowfmindex_amd/libawfm_inter.so -- --mode count --no-e2e
run split${i}_count -- --mode count --no-e2e
done
