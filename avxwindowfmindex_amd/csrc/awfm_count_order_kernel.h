/*
 * awfm_count_order_kernel.h -- counts of a dense-hit batch back from search order to k-mer order in whole lines (round 6).
 *
 * awfmGpuSearchHits with counts only (what awFmParallelSearchCount asks for: ref src/AwFmParallelSearch.c:131-190) on a
 * batch most of whose k-mers survive the deeper table runs orderedSearchKernel, which meets the k-mers in seed order: a store
 * of 4 bytes at counts[k-mer number] from there is a partial line per k-mer -- 3.2 GB of memory traffic for 0.4 GB of
 * counts, 2.2 of the kernel's 7.2 ms on a genome-shaped text (round 5).  Instead the kernel leaves {k-mer number, count}
 * at the k-mer's place in the ORDER (8-byte stores that fill lines as the order is walked), and two passes take the records
 * home:
 *   countScatterKernel  the records into the bucket of their k-mer number's leading bits (each bucket a fixed stretch of the
 *                       output: a bucket of 2^shift numbers never holds more records), a tile of 16384 records at a time: the
 *                       tile's histogram in LDS, one reservation per bucket and tile, then the records again (out of the L2)
 *                       into runs of ~40 -- whole lines but for a run's two ends;
 *   countPlaceKernel    a bucket's records to counts[number]: the bucket's counts are 1 MB, the workgroups of a bucket run on
 *                       ONE XCD (blockIdx % 8: the dispatch order this library's other kernels rely on), two buckets at a time,
 *                       so the 4-byte stores meet in that XCD's L2 and leave it as whole lines.
 * Every k-mer of the order gets its count written, 0 included; the k-mers the order does not hold (ambiguity characters) are
 * the general kernel's, which stores at counts[number] as before.
 */
#ifndef AWFM_COUNT_ORDER_KERNEL_H
#define AWFM_COUNT_ORDER_KERNEL_H

#include "awfm_device.h"

namespace {

constexpr unsigned kCountScatterThreads = 256, kCountScatterPerThread = 64, kCountScatterTile = kCountScatterThreads * kCountScatterPerThread;
constexpr unsigned kCountBucketsMax = 512;
constexpr unsigned kCountPlaceParts = 128; /* workgroups per bucket in countPlaceKernel */

/* the shift that leaves at most kCountBucketsMax buckets of k-mer numbers below n, and windows of counts an L2 holds twice */
inline unsigned countOrderShift(unsigned long long n) {
  unsigned shift = 18; /* 2^18 numbers: 1 MB of counts */
  while (((n - 1ull) >> shift) + 1ull > kCountBucketsMax) shift++;
  return shift;
}

__global__ void __launch_bounds__(kCountScatterThreads)
    countScatterKernel(const uint2 *__restrict__ in, const unsigned *__restrict__ total, const unsigned shift, const unsigned buckets,
                       uint2 *__restrict__ out, unsigned *__restrict__ cursors) {
  __shared__ unsigned sHist[kCountBucketsMax], sBase[kCountBucketsMax];
  const unsigned long long n = *total;
  const unsigned long long tiles = (n + kCountScatterTile - 1ull) / kCountScatterTile;
  for (unsigned long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const unsigned long long first = tile * kCountScatterTile;
    for (unsigned b = threadIdx.x; b < buckets; b += kCountScatterThreads) sHist[b] = 0u;
    __syncthreads();
#pragma unroll 8
    for (unsigned j = 0; j < kCountScatterPerThread; j++) {
      const unsigned long long at = first + (unsigned long long)j * kCountScatterThreads + threadIdx.x;
      if (at < n) atomicAdd(&sHist[in[at].x >> shift], 1u);
    }
    __syncthreads();
    for (unsigned b = threadIdx.x; b < buckets; b += kCountScatterThreads) {
      const unsigned mine = sHist[b];
      sBase[b] = mine ? atomicAdd(&cursors[b], mine) : 0u;
      sHist[b] = 0u;
    }
    __syncthreads();
#pragma unroll 8
    for (unsigned j = 0; j < kCountScatterPerThread; j++) {
      const unsigned long long at = first + (unsigned long long)j * kCountScatterThreads + threadIdx.x;
      if (at < n) {
        const uint2 rec = in[at];
        const unsigned b = rec.x >> shift;
        out[((unsigned long long)b << shift) + sBase[b] + atomicAdd(&sHist[b], 1u)] = rec;
      }
    }
    __syncthreads();
  }
}

/* workgroup id -> (bucket, part): the 8 buckets of a group of buckets side by side (bucket % 8 = blockIdx % 8 = the XCD), the
 * parts of a bucket one after the other on that XCD */
__global__ void __launch_bounds__(256)
    countPlaceKernel(const uint2 *__restrict__ recs, const unsigned *__restrict__ cursors, const unsigned shift, const unsigned buckets,
                     unsigned *__restrict__ counts) {
  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const unsigned bucket = (slot / kCountPlaceParts) * 8u + xcd, part = slot % kCountPlaceParts;
  if (bucket >= buckets) return;
  const unsigned have = cursors[bucket];
  const unsigned per = (have + kCountPlaceParts - 1u) / kCountPlaceParts;
  const unsigned from = part * per, to = from + per < have ? from + per : have;
  const uint2 *mine = recs + ((unsigned long long)bucket << shift);
  for (unsigned i = from + threadIdx.x; i < to; i += 256u) {
    const uint2 rec = mine[i];
    counts[rec.x] = rec.y;
  }
}

}  // namespace

#endif
