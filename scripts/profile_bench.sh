#!/bin/bash
# rocprofv3 runs of bench.py on the GPU box: kernel trace + stats, then PMC passes (each its own run, --pmc only).
# usage: scripts/profile_bench.sh <tag> [bench args...]        ($PROFILE_PASSES="name name ..." restricts the PMC passes)
set -u
TAG=${1:-r2}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
echo "${AWFM_COMMIT:-unknown}" > "$OUT/commit.txt"   # the commit of the tree this box was given (there is no .git here)
cd /tmp && export TMPDIR=/tmp
KERNELS="earchKernel|countScatter|countPlace|lookupSearch|lookupPrep|listTail|exactLookup|SampleAlive|mixedLookupTally|encodeLookup|encodeRecords|partitionRecords|rankMark|rankBlock|rankPlace|sampleAlive|bucketScanShares|walkKernel|finishKernel|fillNoHitKernel|fillSparseKernel|encodeQueriesKernel|encodeCodes|partitionKernel|bucketScanKernel|segmentSumsKernel|tileOffsetsKernel|radix_sort|onesweep|expandHitsKernel|scanTileKernel|scanReduceKernel|sortKeysKernel|bucketKernel"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --no-cpu --no-e2e --no-secondary --no-shard-proxy --no-dense-form --no-wide "$@" > "$OUT/bench_trace.log" 2>&1
declare -A PASS
PASS[fetch]="FETCH_SIZE"
PASS[write]="WRITE_SIZE"
PASS[l2]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
PASS[sq]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
PASS[sq2]="SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
PASS[ea]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum"
PASS[rdreq]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
PASS[tcp]="TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
PASS[ta]="TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum"
PASS[busy]="TCC_BUSY_avr GRBM_TA_BUSY GRBM_TC_BUSY GRBM_EA_BUSY"
# the default passes are the ones that have always come back; `ta` aborted inside rocprofv3 (signal 6) on this pool and then
# sat in its finalisation until the call's limit: name it (or `busy`) in $PROFILE_PASSES only under a short `timeout`
for N in ${PROFILE_PASSES:-fetch write l2 sq sq2 ea tcp}; do
  [ "$N" = "-" ] && continue  # PROFILE_PASSES=-: the kernel trace only
  rocprofv3 --pmc ${PASS[$N]} --kernel-include-regex "$KERNELS" --output-format csv -d "$OUT/pmc_$N" -- python3 "$ROOT/bench.py" --no-cpu --no-e2e --no-secondary --no-shard-proxy --no-dense-form --no-wide --general-steps 0 --steps 2 --warmup 1 "$@" > "$OUT/bench_pmc_$N.log" 2>&1
  rc=$?
  echo "pass $N: $(find "$OUT/pmc_$N" -name '*counter_collection.csv' | wc -l) csv, rc $rc"
done
for f in $(find "$OUT/trace" -name "*kernel_stats.csv"); do echo "== $f"; head -14 "$f" | cut -c1-200; done
