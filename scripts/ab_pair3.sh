#!/bin/bash
cd "$(dirname "$0")/.."
source scripts/ab_lib.sh
for i in 1 2; do for v in A B C; do
run enc${v}${i}_count AWFM_LIB_PATH=$PWD/avxwindowfmindex_amd/libawfm_$v.so -- --no-e2e --mode count
done; done
