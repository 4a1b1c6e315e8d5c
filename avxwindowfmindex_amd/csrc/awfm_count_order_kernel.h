/*
 * awfm_count_order_kernel.h -- counts of a dense-hit batch back from search order to k-mer order in whole lines (round 6).
 *
 * awfmGpuSearchHits with counts only (what awFmParallelSearchCount asks for: ref src/AwFmParallelSearch.c:131-190) on a
 * batch most of whose k-mers survive the deeper table runs orderedSearchKernel, which meets the k-mers in seed order: a store
 * of 4 bytes at counts[k-mer number] from there is a partial line per k-mer -- 3.2 GB of memory traffic for 0.4 GB of
 * counts, 2.2 of the kernel's 7.2 ms on a genome-shaped text (round 5).  Instead the kernel leaves {k-mer number, count}
 * at the k-mer's place in the ORDER (8-byte stores that fill lines as the order is walked), and two passes take the records
 * home:
 *   countScatterKernel  the records into the bucket of their k-mer number's leading bits -- 2^15 numbers a bucket, each bucket a
 *                       fixed stretch of the output (it never holds more records than numbers) --, a tile of 16384 records at a
 *                       time: the tile's histogram in LDS, one reservation per bucket and tile, then the records again (out of the
 *                       L2) to their places; the few thousand lines being filled stay in the L2s until they are whole;
 *   countPlaceKernel    one workgroup per bucket: the bucket's 2^15 counts (128 KB) are put together in LDS and stored as whole
 *                       lines.  (First built with buckets of 2^18 numbers and the 4-byte stores meeting in one XCD's L2: 0.94 ms
 *                       per 10^8 against 0.65 for the scatter -- a store per record is a request per record whatever the L2 merges.)
 * Every k-mer of the order gets its count written, 0 included; what the order does not hold keeps what it has: the pre-fill's 0,
 * the counts a lookup kernel in front of the ordered search stored for the k-mers it searched itself, and -- stored after
 * these passes -- the general kernel's for the k-mers with ambiguity characters.  Batches of up to 2^27 k-mers
 * (4096 buckets); larger ones keep the direct stores.
 */
#ifndef AWFM_COUNT_ORDER_KERNEL_H
#define AWFM_COUNT_ORDER_KERNEL_H

#include "awfm_device.h"

namespace {

constexpr unsigned kCountScatterThreads = 256, kCountScatterPerThread = 64, kCountScatterTile = kCountScatterThreads * kCountScatterPerThread;
constexpr unsigned kCountShift = 15, kCountBucketsMax = 4096, kCountPlaceThreads = 1024;

/* buckets of k-mer numbers below n; 0: more than the passes take */
inline unsigned countOrderBuckets(unsigned long long n) {
  const unsigned long long buckets = ((n - 1ull) >> kCountShift) + 1ull;
  return n != 0ull && buckets <= kCountBucketsMax ? (unsigned)buckets : 0u;
}

__global__ void __launch_bounds__(kCountScatterThreads)
    countScatterKernel(const uint2 *__restrict__ in, const unsigned *__restrict__ total, const unsigned buckets, uint2 *__restrict__ out,
                       unsigned *__restrict__ cursors) {
  __shared__ unsigned sNext[kCountBucketsMax]; /* a tile's records per bucket, then where the bucket's next record goes */
  const unsigned long long n = *total;
  const unsigned long long tiles = (n + kCountScatterTile - 1ull) / kCountScatterTile;
  for (unsigned long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const unsigned long long first = tile * kCountScatterTile;
    for (unsigned b = threadIdx.x; b < buckets; b += kCountScatterThreads) sNext[b] = 0u;
    __syncthreads();
#pragma unroll 8
    for (unsigned j = 0; j < kCountScatterPerThread; j++) {
      const unsigned long long at = first + (unsigned long long)j * kCountScatterThreads + threadIdx.x;
      if (at < n) atomicAdd(&sNext[in[at].x >> kCountShift], 1u);
    }
    __syncthreads();
    for (unsigned b = threadIdx.x; b < buckets; b += kCountScatterThreads) {
      const unsigned mine = sNext[b];
      if (mine) sNext[b] = atomicAdd(&cursors[b], mine); /* one reservation per bucket and tile */
    }
    __syncthreads();
#pragma unroll 8
    for (unsigned j = 0; j < kCountScatterPerThread; j++) {
      const unsigned long long at = first + (unsigned long long)j * kCountScatterThreads + threadIdx.x;
      if (at < n) {
        const uint2 rec = in[at];
        const unsigned b = rec.x >> kCountShift;
        out[((unsigned long long)b << kCountShift) + atomicAdd(&sNext[b], 1u)] = rec;
      }
    }
    __syncthreads();
  }
}

/* one workgroup per bucket: the window of its counts as they stand (the pre-fill's zeros, and whatever a lookup kernel in front
 * of the ordered search has stored for the k-mers it searched itself) -> the bucket's records laid over it in LDS -> back in
 * whole lines.  A bucket without records (the batch went to the lookup kernel alone) is left as it is. */
__global__ void __launch_bounds__(kCountPlaceThreads)
    countPlaceKernel(const uint2 *__restrict__ recs, const unsigned *__restrict__ cursors, const unsigned long long numQueries,
                     unsigned *__restrict__ counts) {
  extern __shared__ unsigned sWindow[]; /* 2^kCountShift words */
  constexpr unsigned kWindow = 1u << kCountShift;
  const unsigned bucket = blockIdx.x;
  const unsigned have = cursors[bucket];
  if (have == 0u) return; /* uniform */
  const unsigned long long firstNumber = (unsigned long long)bucket << kCountShift;
  const unsigned long long left = numQueries - firstNumber;
  const unsigned valid = left < kWindow ? (unsigned)left : kWindow;
  unsigned *to = counts + firstNumber;
  const bool aligned = ((unsigned long long)to & 15ull) == 0ull; /* (a window is 128 KB: aligned whenever counts is) */
  const unsigned vecs = aligned ? valid / 4u : 0u;
  for (unsigned i = threadIdx.x; i < vecs; i += kCountPlaceThreads) {
    const uint4 v = ((const uint4 *)to)[i];
    sWindow[4u * i] = v.x;
    sWindow[4u * i + 1u] = v.y;
    sWindow[4u * i + 2u] = v.z;
    sWindow[4u * i + 3u] = v.w;
  }
  for (unsigned i = vecs * 4u + threadIdx.x; i < valid; i += kCountPlaceThreads) sWindow[i] = to[i];
  __syncthreads();
  const uint2 *mine = recs + firstNumber;
  for (unsigned i = threadIdx.x; i < have; i += kCountPlaceThreads) {
    const uint2 rec = mine[i];
    sWindow[rec.x & (kWindow - 1u)] = rec.y;
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < vecs; i += kCountPlaceThreads)
    ((uint4 *)to)[i] = make_uint4(sWindow[4u * i], sWindow[4u * i + 1u], sWindow[4u * i + 2u], sWindow[4u * i + 3u]);
  for (unsigned i = vecs * 4u + threadIdx.x; i < valid; i += kCountPlaceThreads) to[i] = sWindow[i];
}

}  // namespace

#endif
