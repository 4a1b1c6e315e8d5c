/*
 * awfm_gpu_locate.hip -- from ranges to positions: the exclusive scans behind the hit offsets (scanReduceKernel / scanTileKernel /
 * scanSmallKernel), the expansion of hit windows (expandHitsKernel, expandLongKernel; through the full suffix array the whole
 * locate), the tail of a list-form step in one launch (listTailKernel), and the launches of the LF walk + sampled-SA read
 * (awfm_locate_kernel.h).  ref src/AwFmParallelSearch.c:315-387, src/AwFmSuffixArray.c:114-142, :179-203.
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "awfm_device.h"
#include "awfm_search_kernel.h"
#include "awfm_locate_kernel.h"

namespace {

/* ------------------------------------------------------------------ locate kernels */

constexpr int kScanThreads = 256;
constexpr int kScanItems = 4;
constexpr int kScanTile = kScanThreads * kScanItems;

/* per-tile sums */
/* element i of a scan input: a plain u64 array (SOURCE 0), the length of range i (1; ref
 * src/AwFmIndexStruct.c:126-130), or a u32 count (2) */
constexpr int kScanU64 = 0, kScanRanges = 1, kScanU32 = 2, kScanFlags = 3; /* 3: 1 where a u32 count is not 0 */
template <int SOURCE>
__device__ __forceinline__ unsigned long long scanInput(const void *in, unsigned long long i) {
  if (SOURCE == kScanRanges) {
    const ulonglong2 r = ((const ulonglong2 *)in)[i];
    return r.x <= r.y ? r.y - r.x + 1ull : 0ull;
  }
  if (SOURCE == kScanU32) return ((const unsigned *)in)[i];
  if (SOURCE == kScanFlags) return ((const unsigned *)in)[i] != 0u ? 1ull : 0ull;
  return ((const unsigned long long *)in)[i];
}

template <int SOURCE>
__global__ void __launch_bounds__(kScanThreads)
    scanReduceKernel(const void *__restrict__ in, unsigned long long n,
                     unsigned long long *__restrict__ tileSums) {
  __shared__ unsigned long long sWave[kScanThreads / 64];
  const unsigned long long base = (unsigned long long)blockIdx.x * kScanTile;
  unsigned long long v = 0;
  for (int k = 0; k < kScanItems; k++) {
    const unsigned long long i = base + (unsigned long long)k * kScanThreads + threadIdx.x;
    if (i < n) v += scanInput<SOURCE>(in, i);
  }
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  if ((threadIdx.x & 63) == 0) sWave[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < kScanThreads / 64; w++) t += sWave[w];
    tileSums[blockIdx.x] = t;
  }
}

/* exclusive scan of one tile given the tile's offset (tileOffsets may be NULL for a single tile);
 * also writes the grand total to out[n] when writeTotal */
template <int SOURCE>
__global__ void __launch_bounds__(kScanThreads)
    scanTileKernel(const void *__restrict__ in, unsigned long long n,
                   const unsigned long long *__restrict__ tileOffsets, unsigned long long *__restrict__ out,
                   int writeTotal) {
  __shared__ unsigned long long sWave[kScanThreads / 64];
  const unsigned long long base = (unsigned long long)blockIdx.x * kScanTile + (unsigned long long)threadIdx.x * kScanItems;
  unsigned long long vals[kScanItems];
  unsigned long long sum = 0;
  for (int k = 0; k < kScanItems; k++) {
    vals[k] = base + k < n ? scanInput<SOURCE>(in, base + k) : 0ull;
    sum += vals[k];
  }
  /* inclusive scan of the per-thread sums inside the wave */
  unsigned long long incl = sum;
  const unsigned lane = threadIdx.x & 63u;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long up = __shfl_up(incl, d, 64);
    if (lane >= (unsigned)d) incl += up;
  }
  if (lane == 63u) sWave[threadIdx.x >> 6] = incl;
  __syncthreads();
  unsigned long long waveOffset = 0;
  for (unsigned w = 0; w < (threadIdx.x >> 6); w++) waveOffset += sWave[w];
  unsigned long long running = (tileOffsets ? tileOffsets[blockIdx.x] : 0ull) + waveOffset + incl - sum;
  for (int k = 0; k < kScanItems; k++) {
    if (base + k < n) out[base + k] = running;
    running += vals[k];
  }
  if (writeTotal && base <= n - 1 && n - 1 < base + kScanItems) out[n] = running;
}

/* the same scan of up to kScanSmall elements by ONE workgroup in one launch (16 consecutive elements per thread): the
 * list of hits of a small batch -- 10^4 entries -- is not worth the three launches of the tiled scan (reduce, scan of the
 * sums, tiles: 14 us of a 0.44-ms step) */
constexpr int kScanSmallThreads = 1024, kScanSmallItems = 16, kScanSmall = kScanSmallThreads * kScanSmallItems;
template <int SOURCE>
__global__ void __launch_bounds__(kScanSmallThreads)
    scanSmallKernel(const void *__restrict__ in, unsigned long long n, unsigned long long *__restrict__ out) {
  __shared__ unsigned long long sWave[kScanSmallThreads / 64];
  const unsigned long long base = (unsigned long long)threadIdx.x * kScanSmallItems;
  unsigned long long vals[kScanSmallItems];
  unsigned long long sum = 0;
#pragma unroll
  for (int k = 0; k < kScanSmallItems; k++) {
    vals[k] = base + k < n ? scanInput<SOURCE>(in, base + k) : 0ull;
    sum += vals[k];
  }
  unsigned long long incl = sum;
  const unsigned lane = threadIdx.x & 63u;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long up = __shfl_up(incl, d, 64);
    if (lane >= (unsigned)d) incl += up;
  }
  if (lane == 63u) sWave[threadIdx.x >> 6] = incl;
  __syncthreads();
  unsigned long long running = incl - sum, total = 0;
  for (unsigned w = 0; w < kScanSmallThreads / 64; w++) {
    running += w < (threadIdx.x >> 6) ? sWave[w] : 0ull;
    total += sWave[w];
  }
#pragma unroll
  for (int k = 0; k < kScanSmallItems; k++) {
    if (base + k < n) out[base + k] = running;
    running += vals[k];
  }
  if (threadIdx.x == 0) out[n] = total;
}

/* blocks of 256 threads for n elements, at most 2^22 of them */
inline unsigned cappedGrid(unsigned long long n) {
  const unsigned long long blocks = (n + 255ull) / 256ull;
  return (unsigned)(blocks < (1ull << 22) ? (blocks ? blocks : 1ull) : (1ull << 22));
}

/* positions[hitOffsets[i] + h - hitBegin] = sp_i + h (the BWT positions to trace back) for the hits whose number
 * hitOffsets[i] + h lies in the window [hitBegin, hitEnd), over the queries firstQuery .. firstQuery + n - 1.  The whole
 * batch is the window [0, total) over all queries; a budgeted locate takes the hit list window by window (a window may
 * start and end inside the list of one k-mer). */
/* DENSE: the image carries the full suffix array, so a hit's text position is one read away: positions[...] = dense[sp_i + h]
 * at once, instead of the BWT position for a gather kernel behind this one (10^8 planted 21-mers: a launch and 1.6 GB of
 * intermediate positions written and read back less, 9.7 -> 9.4 ms per step) */
template <bool DENSE>
__global__ void expandHitsKernel(const ulonglong2 *__restrict__ ranges, const unsigned long long *__restrict__ hitOffsets,
                                 unsigned long long firstQuery, unsigned long long n, unsigned long long hitBegin,
                                 unsigned long long hitEnd, unsigned long long *__restrict__ positions,
                                 const DenseSa dense = DenseSa()) {
  /* one wave per 64 queries: short lists by their own lane, long lists by the whole wave; the grid is capped (a
   * launch holds fewer than 2^32 threads), workgroups stride over the batch */
  const unsigned lane = threadIdx.x & 63u;
  for (unsigned long long base = (unsigned long long)blockIdx.x * blockDim.x; base < n;
       base += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned long long i = base + threadIdx.x;
    unsigned long long start = 0, count = 0, sp = 0;
    if (i < n) {
      const unsigned long long from = hitOffsets[firstQuery + i], to = hitOffsets[firstQuery + i + 1];
      const unsigned long long lo = from > hitBegin ? from : hitBegin, hi = to < hitEnd ? to : hitEnd;
      if (lo < hi) { /* batches with few hits: the ranges are not read at all */
        start = lo - hitBegin;
        count = hi - lo;
        sp = ranges[firstQuery + i].x + (lo - from);
      }
    }
    const bool isLong = count > 32ull;
    if (!isLong)
      for (unsigned long long h = 0; h < count; h++) positions[start + h] = DENSE ? denseSaAt(dense, sp + h) : sp + h;
    unsigned long long longMask = __ballot(isLong);
    while (longMask) {
      const int src = __ffsll((long long)longMask) - 1;
      longMask &= longMask - 1ull;
      const unsigned long long s = __shfl(start, src, 64), c = __shfl(count, src, 64), p = __shfl(sp, src, 64);
      for (unsigned long long h = lane; h < c; h += 64ull) positions[s + h] = DENSE ? denseSaAt(dense, p + h) : p + h;
    }
  }
}
/* The same for windows of LONG hit lists (a window of 2^28 hits of 8..11-mers is a few thousand k-mers, 5 * 10^4 hits each):
 * parallel over the HITS.  A workgroup takes chunks of kLongChunk hits of the window, finds the k-mer the chunk begins in
 * (one binary search over the hit offsets per chunk), and walks the k-mers from there, their offsets and first positions
 * staged 64 at a time: every k-mer's part of the chunk is copied by all 256 threads, positions[h - hitBegin] =
 * dense[sp + h - from].  expandHitsKernel<true> gives a k-mer to a wave, which walks a long list one memory latency at a
 * time (2 * 10^6 mixed 8..30-mers, 5.5 * 10^9 hits: 72 ms); an expansion parallel over the k-mers followed by a gather parallel
 * over the hits moved every position three times (154 GB: 32 ms); this kernel reads 4 (5) and writes 8 bytes per hit. */
constexpr unsigned kLongChunk = 16384, kLongStage = 64;
__global__ void __launch_bounds__(256)
    expandLongKernel(const ulonglong2 *__restrict__ ranges, const unsigned long long *__restrict__ hitOffsets,
                     const unsigned long long firstQuery, const unsigned long long n, const unsigned long long hitBegin,
                     const unsigned long long hitEnd, unsigned long long *__restrict__ positions, const DenseSa dense) {
  __shared__ unsigned long long sOff[kLongStage + 1], sSp[kLongStage], sFirst;
  const unsigned tid = threadIdx.x;
  const unsigned long long chunks = (hitEnd - hitBegin + kLongChunk - 1ull) / kLongChunk;
  for (unsigned long long chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
    const unsigned long long c0 = hitBegin + chunk * kLongChunk, c1 = c0 + kLongChunk < hitEnd ? c0 + kLongChunk : hitEnd;
    if (tid == 0) { /* the last k-mer of the window whose list begins at or before the chunk (the first one when none does) */
      unsigned long long lo = 0, hi = n;
      while (hi - lo > 1ull) {
        const unsigned long long mid = (lo + hi) >> 1;
        if (hitOffsets[firstQuery + mid] <= c0) lo = mid;
        else hi = mid;
      }
      sFirst = lo;
    }
    __syncthreads();
    bool done = false; /* uniform */
    for (unsigned long long qb = sFirst; qb < n && !done; qb += kLongStage) {
      if (tid <= kLongStage) sOff[tid] = hitOffsets[firstQuery + (qb + tid < n ? qb + tid : n)];
      if (tid < kLongStage) sSp[tid] = qb + tid < n ? ranges[firstQuery + qb + tid].x : 0ull;
      __syncthreads();
      for (unsigned i = 0; i < kLongStage && qb + i < n; i++) {
        const unsigned long long from = sOff[i], to = sOff[i + 1];
        if (from >= c1) {
          done = true;
          break;
        }
        const unsigned long long lo = from > c0 ? from : c0, hi = to < c1 ? to : c1;
        if (lo < hi) {
          const unsigned long long src = sSp[i] + (lo - from) - lo; /* dense[src + h] for hit h */
          unsigned long long h = lo + tid;
          for (; h + 768ull < hi; h += 1024ull) { /* four gathers of the thread in flight */
            const unsigned long long a = denseSaAt(dense, src + h), b = denseSaAt(dense, src + h + 256ull), c = denseSaAt(dense, src + h + 512ull),
                                     d = denseSaAt(dense, src + h + 768ull);
            positions[h - hitBegin] = a;
            positions[h + 256ull - hitBegin] = b;
            positions[h + 512ull - hitBegin] = c;
            positions[h + 768ull - hitBegin] = d;
          }
          for (; h < hi; h += 256ull) positions[h - hitBegin] = denseSaAt(dense, src + h);
        }
      }
      if (sOff[kLongStage] >= c1) done = true;
      __syncthreads(); /* the stage is written again */
    }
    __syncthreads(); /* sFirst is written again */
  }
}

/* ---- the tail of a step whose results are the LIST of the k-mers with hits, in ONE launch (round 5) ----
 * awfmGpuSearchHitsCompact leaves {k-mer number, range} entries in the order the waves appended them; what follows -- the
 * list in k-mer order, the hit offsets over it, the positions -- was a memset, four ranking kernels over a bitmap of the
 * batch, a scan and the expand / gather kernel: seven dependent launches of 5-14 us each, 48 of the 470 us a
 * 1.25 * 10^7-k-mer shard of an 8-GPU run takes (ref src/AwFmParallelSearch.c:315-365 does this per k-mer on the host).
 * Here workgroup c owns the k-mer numbers [c R, (c + 1) R): it reads the whole list once (the keys: 4 bytes an entry, out
 * of the L2), counts the entries below its range and their hits -- its own prefix, no scan across workgroups, no atomics,
 * no scratch --, gathers its own entries in LDS, ranks them by counting, scans their lengths and writes them out: sorted
 * entry, hit offset, and the hits' positions (through the full suffix array when the image has it).  The list's k-mer
 * numbers are distinct, so a range of kListTailSlots numbers holds at most that many entries: a workgroup whose range
 * holds more (clustered hits) goes through it in sub-ranges of that width, re-reading the keys for each.  The last workgroup
 * knows the total and fills what lies behind the list. */
constexpr unsigned kListTailThreads = 1024, kListTailSlots = 2048;
constexpr unsigned kListTailMaxEntries = 1u << 18; /* longer lists: the three calls this kernel replaces (the tail is then no longer launch-bound) */
template <bool DENSE>
__global__ void __launch_bounds__(kListTailThreads)
    listTailKernel(const unsigned *__restrict__ inKmers, const ulonglong2 *__restrict__ inRanges, const unsigned *__restrict__ count,
                   const unsigned cap, const unsigned long long numQueries, unsigned *__restrict__ outKmers,
                   ulonglong2 *__restrict__ outRanges, unsigned long long *__restrict__ hitOffsets, const unsigned long long capacityHits,
                   unsigned long long *__restrict__ positions, const DenseSa dense) {
  __shared__ unsigned sKey[kListTailSlots], sOrder[kListTailSlots];
  __shared__ ulonglong2 sRange[kListTailSlots];
  __shared__ unsigned long long sWave[kListTailThreads / 64], sRed[2][kListTailThreads / 64];
  __shared__ unsigned sMine;
  /* entries with more than kHuge hits (a random k-mer that falls into a repeat family of a genome-shaped text: 10^5) are
   * expanded by the whole workgroup, four gathers a thread in flight -- by one wave, 64 hits a trip, such an entry was a
   * chain of 1500 memory latencies, and 0.2 ms of a 0.57-ms shard step on that text */
  constexpr unsigned kHugeSlots = 64;
  constexpr unsigned long long kHuge = 4096;
  __shared__ unsigned long long sHugeOff[kHugeSlots], sHugeCount[kHugeSlots], sHugeFrom[kHugeSlots];
  __shared__ unsigned sHugeN;
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned n = *count < cap ? *count : cap;
  const unsigned long long width = (numQueries + gridDim.x - 1ull) / gridDim.x;
  const unsigned long long lo = width * blockIdx.x;
  const bool lastGroup = blockIdx.x == gridDim.x - 1u;
  const unsigned long long hi = lastGroup ? (1ull << 32) : lo + width; /* (a number that is no k-mer of the batch sorts last) */
  if (tid == 0) sMine = 0u;
  __syncthreads();
  /* one pass over the list: entries below the range (count, hits), own entries into the slots */
  /* Sixteen entries a thread and trip -- two groups of eight --, all their loads requested before any is used, and requested
   * before the list's length has arrived (the trips run over the list's CAPACITY, which is an argument; an entry beyond the
   * length is read and ignored): the pass is a chain of memory latencies, 2 us each -- one entry a trip took 13 us over the
   * 9 * 10^3 entries of a shard's list and would take 100 over the 7 * 10^4 of the whole batch's.  (Measured and dropped: one
   * slot reservation per wave -- a shuffle scan of the threads' counts and a second pass over the keys -- instead of an LDS
   * atomic per own entry: 20-21 us against 17 for a shard's list; and with it 128 / 64 / 32 workgroups instead of 256: 21 / 23 /
   * 27 us -- fewer workgroups re-read less of the list and are no faster.) */
  unsigned long long below = 0, belowHits = 0;
  constexpr unsigned kPer = 8, kGroups = 2;
  const bool vec = ((unsigned long long)inKmers & 15ull) == 0ull;
  const bool allRanges = cap <= 16384u; /* a short list: every range is requested beside its key, not behind it */
  for (unsigned trip = 0; trip < cap; trip += kListTailThreads * kPer * kGroups) {
    unsigned key32[kGroups][kPer];
    ulonglong2 r[kGroups][kPer];
#pragma unroll
    for (unsigned u = 0; u < kGroups; u++) {
      const unsigned base = trip + u * kListTailThreads * kPer + tid * kPer;
      if (vec && base + kPer <= cap) {
        const uint4 a = *(const uint4 *)(inKmers + base), b = *(const uint4 *)(inKmers + base + 4u);
        key32[u][0] = a.x, key32[u][1] = a.y, key32[u][2] = a.z, key32[u][3] = a.w;
        key32[u][4] = b.x, key32[u][5] = b.y, key32[u][6] = b.z, key32[u][7] = b.w;
      } else {
#pragma unroll
        for (unsigned j = 0; j < kPer; j++) key32[u][j] = base + j < cap ? inKmers[base + j] : 0xFFFFFFFFu;
      }
    }
#pragma unroll
    for (unsigned u = 0; u < kGroups; u++) {
      const unsigned base = trip + u * kListTailThreads * kPer + tid * kPer;
#pragma unroll
      for (unsigned j = 0; j < kPer; j++)
        r[u][j] = base + j < cap && (allRanges || (unsigned long long)key32[u][j] < hi) ? inRanges[base + j] : make_ulonglong2(1ull, 0ull);
    }
#pragma unroll
    for (unsigned u = 0; u < kGroups; u++) {
      const unsigned base = trip + u * kListTailThreads * kPer + tid * kPer;
#pragma unroll
      for (unsigned j = 0; j < kPer; j++) {
        const unsigned long long key = base + j < n ? (unsigned long long)key32[u][j] : ~0ull;
        if (key < lo) {
          below++;
          belowHits += r[u][j].x <= r[u][j].y ? r[u][j].y - r[u][j].x + 1ull : 0ull;
        } else if (key < hi) {
          const unsigned at = atomicAdd(&sMine, 1u);
          if (at < kListTailSlots) {
            sKey[at] = (unsigned)key;
            sRange[at] = r[u][j];
          }
        }
      }
    }
  }
  for (int d = 32; d >= 1; d >>= 1) {
    below += __shfl_xor(below, d, 64);
    belowHits += __shfl_xor(belowHits, d, 64);
  }
  if (lane == 0) {
    sRed[0][wave] = below;
    sRed[1][wave] = belowHits;
  }
  __syncthreads();
  unsigned long long rankBase = 0, hitBase = 0; /* uniform: where the next entry of this workgroup goes */
  for (unsigned v = 0; v < kListTailThreads / 64; v++) {
    rankBase += sRed[0][v];
    hitBase += sRed[1][v];
  }
  const unsigned mine = sMine;
  /* the entries in the slots [0, m): ranked, scanned, written out */
  auto emit = [&](const unsigned m) {
    for (unsigned j = tid; j < m; j += kListTailThreads) {
      const unsigned key = sKey[j];
      unsigned r = 0; /* (ties -- a k-mer listed twice, which a search never does -- by slot: the ranks stay a permutation) */
      for (unsigned i = 0; i < m; i++) r += sKey[i] < key || (sKey[i] == key && i < j) ? 1u : 0u;
      sOrder[r] = j;
    }
    __syncthreads();
    for (unsigned base = 0; base < m; base += kListTailThreads) { /* uniform trip count */
      const unsigned r = base + tid;
      unsigned key = 0;
      ulonglong2 range = make_ulonglong2(1ull, 0ull);
      unsigned long long len = 0;
      if (r < m) {
        const unsigned j = sOrder[r];
        key = sKey[j];
        range = sRange[j];
        len = range.x <= range.y ? range.y - range.x + 1ull : 0ull;
      }
      unsigned long long incl = len;
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long up = __shfl_up(incl, d, 64);
        if (lane >= (unsigned)d) incl += up;
      }
      if (lane == 63u) sWave[wave] = incl;
      if (tid == 0) sHugeN = 0u;
      __syncthreads();
      unsigned long long before = hitBase, chunk = 0;
      for (unsigned v = 0; v < kListTailThreads / 64; v++) {
        before += v < wave ? sWave[v] : 0ull;
        chunk += sWave[v];
      }
      const unsigned long long off = before + incl - len;
      if (r < m) {
        outKmers[rankBase + r] = key;
        outRanges[rankBase + r] = range;
        hitOffsets[rankBase + r] = off;
      }
      /* the hits of the entry: short lists by their own lane, long ones by the wave (as expandHitsKernel) */
      unsigned long long countHere = 0;
      if (positions && off < capacityHits) countHere = off + len <= capacityHits ? len : capacityHits - off;
      bool isHuge = countHere > kHuge;
      if (isHuge) {
        const unsigned at = atomicAdd(&sHugeN, 1u);
        if (at < kHugeSlots) {
          sHugeOff[at] = off;
          sHugeCount[at] = countHere;
          sHugeFrom[at] = range.x;
        } else {
          isHuge = false; /* (more than the slots hold in one trip: by its wave, below) */
        }
      }
      const bool isLong = countHere > 32ull && !isHuge;
      if (!isLong && !isHuge)
        for (unsigned long long h = 0; h < countHere; h++) positions[off + h] = DENSE ? denseSaAt(dense, range.x + h) : range.x + h;
      unsigned long long longMask = __ballot(isLong);
      while (longMask) {
        const int src = __ffsll((long long)longMask) - 1;
        longMask &= longMask - 1ull;
        const unsigned long long o = __shfl(off, src, 64), c = __shfl(countHere, src, 64), p = __shfl(range.x, src, 64);
        for (unsigned long long h = lane; h < c; h += 64ull) positions[o + h] = DENSE ? denseSaAt(dense, p + h) : p + h;
      }
      hitBase += chunk;
      __syncthreads(); /* sWave is written again */
      const unsigned huge = sHugeN < kHugeSlots ? sHugeN : kHugeSlots; /* uniform */
      for (unsigned e = 0; e < huge; e++) {
        const unsigned long long o = sHugeOff[e], c = sHugeCount[e], p = sHugeFrom[e];
        unsigned long long h = tid;
        for (; h + 3ull * kListTailThreads < c; h += 4ull * kListTailThreads) {
          unsigned long long v[4];
#pragma unroll
          for (unsigned u = 0; u < 4u; u++) v[u] = DENSE ? denseSaAt(dense, p + h + u * kListTailThreads) : p + h + u * kListTailThreads;
#pragma unroll
          for (unsigned u = 0; u < 4u; u++) positions[o + h + u * kListTailThreads] = v[u];
        }
        for (; h < c; h += kListTailThreads) positions[o + h] = DENSE ? denseSaAt(dense, p + h) : p + h;
      }
      if (huge) __syncthreads(); /* the slots are written again */
    }
    rankBase += m;
  };
  if (mine <= kListTailSlots) {
    emit(mine);
  } else {
    /* more entries than slots in this range: sub-ranges of kListTailSlots k-mer numbers, the keys read again for each */
    for (unsigned long long sub = lo; sub < hi && rankBase < n; sub += kListTailSlots) {
      const unsigned long long subEnd = sub + kListTailSlots < hi ? sub + kListTailSlots : hi;
      __syncthreads();
      if (tid == 0) sMine = 0u;
      __syncthreads();
      for (unsigned i = tid; i < n; i += kListTailThreads) {
        const unsigned long long key = inKmers[i];
        if (key >= sub && key < subEnd) {
          const unsigned at = atomicAdd(&sMine, 1u);
          if (at < kListTailSlots) { /* (more only when the list names a k-mer twice, which a search never does) */
            sKey[at] = (unsigned)key;
            sRange[at] = inRanges[i];
          }
        }
      }
      __syncthreads();
      const unsigned m = sMine < kListTailSlots ? sMine : kListTailSlots;
      if (m) emit(m);
    }
  }
  if (lastGroup) { /* behind the list: empty entries, every offset the total */
    for (unsigned long long i = rankBase + tid; i <= cap; i += kListTailThreads) {
      if (i < cap) {
        outKmers[i] = 0xFFFFFFFFu;
        outRanges[i] = make_ulonglong2(1ull, 0ull);
      }
      hitOffsets[i] = hitBase;
    }
  }
}
}  // namespace

extern "C" {

/* scratch: tile sums (and their scanned offsets) per level */
uint64_t awfmGpuScanScratchBytes(uint64_t numQueries) {
  uint64_t words = 4;
  uint64_t level = numQueries;
  while (level > (uint64_t)kScanTile) {
    level = (level + kScanTile - 1) / kScanTile;
    words += 2 * level + 2; /* sums + their scanned offsets */
  }
  return words * 8 + 64;
}

namespace {
/* exclusive scan of in[0..n) into out[0..n] (out[n] = total), recursive over tiles */
extern "C++" {
template <int SOURCE>
enum AwFmReturnCode scanRecursive(const void *in, uint64_t n, unsigned long long *out, unsigned long long *scratch,
                                  hipStream_t s) {
  const uint64_t tiles = (n + kScanTile - 1) / kScanTile;
  if (tiles > 1 && n <= (uint64_t)kScanSmall) {
    hipLaunchKernelGGL(scanSmallKernel<SOURCE>, dim3(1), dim3(kScanSmallThreads), 0, s, in, (unsigned long long)n, out);
    AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
    return AwFmSuccess;
  }
  if (tiles <= 1) {
    hipLaunchKernelGGL(scanTileKernel<SOURCE>, dim3(1), dim3(kScanThreads), 0, s, in, (unsigned long long)n,
                       (const unsigned long long *)nullptr, out, 1);
    AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
    return AwFmSuccess;
  }
  unsigned long long *sums = scratch;
  unsigned long long *offs = scratch + tiles;
  hipLaunchKernelGGL(scanReduceKernel<SOURCE>, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, in,
                     (unsigned long long)n, sums);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  const enum AwFmReturnCode rc = scanRecursive<kScanU64>(sums, tiles, offs, scratch + 2 * tiles + 2, s);
  if (rc != AwFmSuccess) return rc;
  hipLaunchKernelGGL(scanTileKernel<SOURCE>, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, in,
                     (unsigned long long)n, (const unsigned long long *)offs, out, 1);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  return AwFmSuccess;
}
}  // extern "C++"
}  // namespace

enum AwFmReturnCode awfmGpuHitOffsets(AwFmGpuIndex *g, const struct AwFmSearchRange *dRanges, uint64_t numQueries,
                                      uint64_t *dHitOffsets, void *dScratch, uint64_t *totalHits, void *stream) {
  if (!g || !dRanges || !dHitOffsets || !dScratch || !totalHits) {
    setError("awfmGpuHitOffsets: null argument");
    return AwFmNullPtrError;
  }
  DeviceGuard guard(g->device);
  hipStream_t s = (hipStream_t)stream;
  *totalHits = 0;
  if (numQueries == 0) {
    AWFM_HIP_TRY(hipMemsetAsync(dHitOffsets, 0, 8, s), AwFmGeneralFailure);
    AWFM_HIP_TRY(hipStreamSynchronize(s), AwFmGeneralFailure);
    return AwFmSuccess;
  }
  /* the scan reads the ranges directly (lengths are formed on the fly).  rocPRIM's one-pass look-back scan over
   * the same input measured 0.65 ms per 10^8 queries against 0.76 ms for these two passes: not worth a dependency */
  const enum AwFmReturnCode rc =
      scanRecursive<kScanRanges>(dRanges, numQueries, (unsigned long long *)dHitOffsets, (unsigned long long *)dScratch, s);
  if (rc != AwFmSuccess) return rc;
  AWFM_HIP_TRY(hipMemcpyAsync(totalHits, dHitOffsets + numQueries, 8, hipMemcpyDeviceToHost, s), AwFmGeneralFailure);
  AWFM_HIP_TRY(hipStreamSynchronize(s), AwFmGeneralFailure);
  return AwFmSuccess;
}

enum AwFmReturnCode awfmGpuHitOffsetsFromCounts(AwFmGpuIndex *g, const uint32_t *dCounts, uint64_t numQueries,
                                                uint64_t *dHitOffsets, void *dScratch, uint64_t *totalHits, void *stream) {
  if (!g || !dCounts || !dHitOffsets || !dScratch || !totalHits) {
    setError("awfmGpuHitOffsetsFromCounts: null argument");
    return AwFmNullPtrError;
  }
  if (g->dev.bwtLength >= (1ull << 32)) {
    setError("awfmGpuHitOffsetsFromCounts: 32-bit counts are exact only for images below 2^32 positions; use awfmGpuHitOffsets");
    return AwFmUnsupportedVersionError;
  }
  DeviceGuard guard(g->device);
  hipStream_t s = (hipStream_t)stream;
  *totalHits = 0;
  if (numQueries == 0) {
    AWFM_HIP_TRY(hipMemsetAsync(dHitOffsets, 0, 8, s), AwFmGeneralFailure);
    AWFM_HIP_TRY(hipStreamSynchronize(s), AwFmGeneralFailure);
    return AwFmSuccess;
  }
  const enum AwFmReturnCode rc =
      scanRecursive<kScanU32>(dCounts, numQueries, (unsigned long long *)dHitOffsets, (unsigned long long *)dScratch, s);
  if (rc != AwFmSuccess) return rc;
  AWFM_HIP_TRY(hipMemcpyAsync(totalHits, dHitOffsets + numQueries, 8, hipMemcpyDeviceToHost, s), AwFmGeneralFailure);
  AWFM_HIP_TRY(hipStreamSynchronize(s), AwFmGeneralFailure);
  return AwFmSuccess;
}

}  // extern "C"

enum AwFmReturnCode awfmGpuHitOffsetsAsync(AwFmGpuIndex *g, const uint32_t *dCounts, const struct AwFmSearchRange *dRanges,
                                           uint64_t numQueries, uint64_t *dHitOffsets, void *dScratch,
                                           unsigned long long *pinnedTotal, hipStream_t s) {
  if (!g || (!dCounts && !dRanges) || !dHitOffsets || !dScratch || !pinnedTotal || numQueries == 0) {
    setError("awfmGpuHitOffsetsAsync: null argument");
    return AwFmNullPtrError;
  }
  const enum AwFmReturnCode rc =
      dCounts ? scanRecursive<kScanU32>(dCounts, numQueries, (unsigned long long *)dHitOffsets, (unsigned long long *)dScratch, s)
              : scanRecursive<kScanRanges>(dRanges, numQueries, (unsigned long long *)dHitOffsets, (unsigned long long *)dScratch, s);
  if (rc != AwFmSuccess) return rc;
  AWFM_HIP_TRY(hipMemcpyAsync(pinnedTotal, dHitOffsets + numQueries, 8, hipMemcpyDeviceToHost, s), AwFmGeneralFailure);
  return AwFmSuccess;
}

enum AwFmReturnCode awfmGpuScanFlags(AwFmGpuIndex *g, const uint32_t *dCounts, uint64_t numQueries, uint64_t *dFlagOffsets,
                                     void *dScratch, hipStream_t s) {
  if (!g || !dCounts || !dFlagOffsets || !dScratch || numQueries == 0) {
    setError("awfmGpuScanFlags: null argument");
    return AwFmNullPtrError;
  }
  return scanRecursive<kScanFlags>(dCounts, numQueries, (unsigned long long *)dFlagOffsets, (unsigned long long *)dScratch, s);
}

extern "C" {

enum AwFmReturnCode awfmGpuLocate(AwFmGpuIndex *g, const struct AwFmSearchRange *dRanges,
                                  const uint64_t *dHitOffsets, uint64_t numQueries, uint64_t totalHits,
                                  uint64_t *dPositions, void *stream) {
  return awfmGpuLocateTo(g, dRanges, dHitOffsets, numQueries, totalHits, dPositions, dPositions, stream);
}

enum AwFmReturnCode awfmGpuLocateTo(AwFmGpuIndex *g, const struct AwFmSearchRange *dRanges,
                                    const uint64_t *dHitOffsets, uint64_t numQueries, uint64_t totalHits,
                                    uint64_t *dPositions, uint64_t *outPositions, void *stream) {
  return awfmGpuLocateWindow(g, dRanges, dHitOffsets, 0, numQueries, 0, totalHits, dPositions, outPositions, stream);
}

enum AwFmReturnCode awfmGpuLocateWindow(AwFmGpuIndex *g, const struct AwFmSearchRange *dRanges, const uint64_t *dHitOffsets,
                                        uint64_t queryBegin, uint64_t queryEnd, uint64_t hitBegin, uint64_t hitEnd,
                                        uint64_t *dPositions, uint64_t *outPositions, void *stream) {
  if (!g) {
    setError("awfmGpuLocate: null image");
    return AwFmNullPtrError;
  }
  if (queryEnd <= queryBegin || hitEnd <= hitBegin) return AwFmSuccess;
  if (!dRanges || !dHitOffsets || !dPositions || !outPositions) {
    setError("awfmGpuLocate: null argument");
    return AwFmNullPtrError;
  }
  const uint64_t numQueries = queryEnd - queryBegin, totalHits = hitEnd - hitBegin;
  DeviceGuard guard(g->device);
  hipStream_t s = (hipStream_t)stream;
  if (g->dDenseSa && totalHits < 64ull * numQueries) {
    /* the full suffix array: expand and gather in one kernel, straight to where the positions go */
    hipLaunchKernelGGL(expandHitsKernel<true>, dim3(cappedGrid(numQueries)), dim3(256), 0, s,
                       (const ulonglong2 *)dRanges, (const unsigned long long *)dHitOffsets, (unsigned long long)queryBegin,
                       (unsigned long long)numQueries, (unsigned long long)hitBegin, (unsigned long long)hitEnd,
                       (unsigned long long *)outPositions, denseSaOf(g));
    AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
    return AwFmSuccess;
  }
  if (g->dDenseSa) {
    /* long hit lists (64 hits per k-mer and more on average): parallel over the hits (expandLongKernel; the expansion parallel
     * over the k-mers followed by a gather parallel over the hits moved every position three times: 32 against 15 ms for
     * 2 * 10^6 mixed 8..30-mers with 5.5 * 10^9 hits) */
    const unsigned long long chunks = (totalHits + kLongChunk - 1ull) / kLongChunk, resident = (unsigned long long)g->numCUs * 8ull;
    hipLaunchKernelGGL(expandLongKernel, dim3((unsigned)(chunks < resident ? chunks : resident)), dim3(256), 0, s,
                       (const ulonglong2 *)dRanges, (const unsigned long long *)dHitOffsets, (unsigned long long)queryBegin,
                       (unsigned long long)numQueries, (unsigned long long)hitBegin, (unsigned long long)hitEnd,
                       (unsigned long long *)outPositions, denseSaOf(g));
    AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
    return AwFmSuccess;
  }
  hipLaunchKernelGGL(expandHitsKernel<false>, dim3(cappedGrid(numQueries)), dim3(256), 0, s,
                     (const ulonglong2 *)dRanges, (const unsigned long long *)dHitOffsets, (unsigned long long)queryBegin,
                     (unsigned long long)numQueries, (unsigned long long)hitBegin, (unsigned long long)hitEnd,
                     (unsigned long long *)dPositions, DenseSa());
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  return awfmGpuLaunchLocate(g, totalHits, (unsigned long long *)dPositions, s, (unsigned long long *)outPositions);
}

/* see include/awfm_gpu.h */
enum AwFmReturnCode awfmGpuHitOffsetsOnDevice(AwFmGpuIndex *g, const uint32_t *dCounts, const struct AwFmSearchRange *dRanges,
                                              uint64_t numQueries, uint64_t *dHitOffsets, void *dScratch, void *stream) {
  if (!g || (!dCounts && !dRanges) || !dHitOffsets || !dScratch || numQueries == 0) {
    setError("awfmGpuHitOffsetsOnDevice: null argument");
    return AwFmNullPtrError;
  }
  if (dCounts && g->dev.bwtLength >= (1ull << 32)) {
    setError("awfmGpuHitOffsetsOnDevice: 32-bit counts are exact only for images below 2^32 positions; pass the ranges");
    return AwFmUnsupportedVersionError;
  }
  DeviceGuard guard(g->device);
  hipStream_t s = (hipStream_t)stream;
  return dCounts ? scanRecursive<kScanU32>(dCounts, numQueries, (unsigned long long *)dHitOffsets, (unsigned long long *)dScratch, s)
                 : scanRecursive<kScanRanges>(dRanges, numQueries, (unsigned long long *)dHitOffsets, (unsigned long long *)dScratch, s);
}

enum AwFmReturnCode awfmGpuLocateOnDevice(AwFmGpuIndex *g, const struct AwFmSearchRange *dRanges, const uint64_t *dHitOffsets,
                                          uint64_t numQueries, uint64_t capacityHits, uint64_t *dPositions, void *stream) {
  if (!g || !dRanges || !dHitOffsets || !dPositions) {
    setError("awfmGpuLocateOnDevice: null argument");
    return AwFmNullPtrError;
  }
  if (numQueries == 0 || capacityHits == 0) return AwFmSuccess;
  DeviceGuard guard(g->device);
  hipStream_t s = (hipStream_t)stream;
  /* the window [0, capacity) of the hit list: the hits beyond what `dPositions` holds are left out (the caller sees from
   * the total, when it gets to read it, that the buffer was too small) */
  if (g->dDenseSa) { /* the full suffix array: expand and gather in one kernel */
    hipLaunchKernelGGL(expandHitsKernel<true>, dim3(cappedGrid(numQueries)), dim3(256), 0, s, (const ulonglong2 *)dRanges,
                       (const unsigned long long *)dHitOffsets, 0ull, (unsigned long long)numQueries, 0ull,
                       (unsigned long long)capacityHits, (unsigned long long *)dPositions, denseSaOf(g));
    AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
    return AwFmSuccess;
  }
  hipLaunchKernelGGL(expandHitsKernel<false>, dim3(cappedGrid(numQueries)), dim3(256), 0, s, (const ulonglong2 *)dRanges,
                     (const unsigned long long *)dHitOffsets, 0ull, (unsigned long long)numQueries, 0ull,
                     (unsigned long long)capacityHits, (unsigned long long *)dPositions, DenseSa());
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  const unsigned long long *total = (const unsigned long long *)dHitOffsets + numQueries;
  return awfmGpuLaunchLocate(g, capacityHits, (unsigned long long *)dPositions, s, (unsigned long long *)dPositions, total);
}


/* see include/awfm_gpu.h */
enum AwFmReturnCode awfmGpuListLocateOnDevice(AwFmGpuIndex *g, const uint32_t *dHitKmers, const struct AwFmSearchRange *dHitRanges,
                                              uint32_t capacity, const uint32_t *dNumHits, uint64_t numQueries, uint32_t *dSortedKmers,
                                              struct AwFmSearchRange *dSortedRanges, uint64_t *dHitOffsets, uint64_t capacityHits,
                                              uint64_t *dPositions, void *stream) {
  if (!g || !dHitKmers || !dHitRanges || !dNumHits || !dSortedKmers || !dSortedRanges || !dHitOffsets) {
    setError("awfmGpuListLocateOnDevice: null argument");
    return AwFmNullPtrError;
  }
  if (capacity == 0 || numQueries == 0 || numQueries >= 0xFFFFFFFFull) {
    setError("awfmGpuListLocateOnDevice: a list needs a capacity and a batch of 1 .. 2^32 - 2 k-mers");
    return AwFmIllegalPositionError;
  }
  if ((const void *)dHitKmers == (const void *)dSortedKmers || (const void *)dHitRanges == (const void *)dSortedRanges) {
    setError("awfmGpuListLocateOnDevice: the list is not put in order in place (awfmGpuSortHitsOnDevice does that)");
    return AwFmIllegalPositionError;
  }
  DeviceGuard guard(g->device);
  hipStream_t s = (hipStream_t)stream;
  if (!dPositions) capacityHits = 0;
  if (capacity > kListTailMaxEntries) {
    /* a long list: copy, rank in a bitmap of the batch, scan, expand (what a caller did before this entry point existed) */
    AWFM_HIP_TRY(hipMemcpyAsync(dSortedKmers, dHitKmers, (size_t)capacity * 4u, hipMemcpyDeviceToDevice, s), AwFmGeneralFailure);
    AWFM_HIP_TRY(hipMemcpyAsync(dSortedRanges, dHitRanges, (size_t)capacity * 16u, hipMemcpyDeviceToDevice, s), AwFmGeneralFailure);
    enum AwFmReturnCode rc = awfmGpuSortHitsOnDevice(g, dSortedKmers, dSortedRanges, capacity, dNumHits, numQueries, stream);
    if (rc != AwFmSuccess) return rc;
    /* the scan's scratch is this call's own, allocated and freed in stream order: the image's grow-only work buffer belongs
     * to the host-buffer entry points, which hold its mutex for their whole synchronous call and may re-allocate it (advisor,
     * round 5: handing it out beyond that mutex let a concurrent host call overwrite or free the scratch of a scan in flight) */
    void *scratch = nullptr;
    AWFM_HIP_TRY(hipMallocAsync(&scratch, awfmGpuScanScratchBytes(capacity), s), AwFmAllocationFailure);
    rc = awfmGpuHitOffsetsOnDevice(g, nullptr, dSortedRanges, capacity, dHitOffsets, scratch, stream);
    const hipError_t freed = hipFreeAsync(scratch, s);
    if (rc != AwFmSuccess) return rc;
    AWFM_HIP_TRY(freed, AwFmGeneralFailure);
    if (capacityHits == 0) return rc;
    return awfmGpuLocateOnDevice(g, dSortedRanges, dHitOffsets, capacity, capacityHits, dPositions, stream);
  }
  /* a workgroup per stretch of the batch; a short list does not need the whole chip */
  unsigned grid = capacity / 16u;
  grid = grid < 1u ? 1u : (grid > (unsigned)g->numCUs ? (unsigned)g->numCUs : grid);
  if ((unsigned long long)grid > numQueries) grid = (unsigned)numQueries;
  if (g->dDenseSa)
    hipLaunchKernelGGL(listTailKernel<true>, dim3(grid), dim3(kListTailThreads), 0, s, (const unsigned *)dHitKmers, (const ulonglong2 *)dHitRanges,
                       (const unsigned *)dNumHits, (unsigned)capacity, (unsigned long long)numQueries, (unsigned *)dSortedKmers,
                       (ulonglong2 *)dSortedRanges, (unsigned long long *)dHitOffsets, (unsigned long long)capacityHits,
                       (unsigned long long *)dPositions, denseSaOf(g));
  else
    hipLaunchKernelGGL(listTailKernel<false>, dim3(grid), dim3(kListTailThreads), 0, s, (const unsigned *)dHitKmers, (const ulonglong2 *)dHitRanges,
                       (const unsigned *)dNumHits, (unsigned)capacity, (unsigned long long)numQueries, (unsigned *)dSortedKmers,
                       (ulonglong2 *)dSortedRanges, (unsigned long long *)dHitOffsets, (unsigned long long)capacityHits,
                       (unsigned long long *)dPositions, DenseSa());
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  if (g->dDenseSa || capacityHits == 0) return AwFmSuccess;
  /* no full suffix array: the kernel left the BWT positions; the LF walk and the sample reads take them from there */
  return awfmGpuLaunchLocate(g, capacityHits, (unsigned long long *)dPositions, s, (unsigned long long *)dPositions,
                      (const unsigned long long *)dHitOffsets + capacity);
}

}  // extern "C"

/* LF-walk + sampled-SA kernel over `totalHits` BWT positions stored in dPositions (in place) */
enum AwFmReturnCode awfmGpuLaunchLocate(AwFmGpuIndex *g, unsigned long long totalHits, unsigned long long *dPositions,
                                        hipStream_t s, unsigned long long *out, const unsigned long long *totalOnDevice, unsigned stepCap) {
  {
    /* the walk runs at the rate the chip delivers random granules whatever the group width (17.2 / 17.5 / 18.3 ms
     * for g4 / g2 / g1 on 1.0007*10^8 hits); four lanes keep the fewest instructions per step */
    int lanes = g->kernel == AWFM_GPU_KERNEL_AUTO ? 4 : awfmGpuLanesPerQuery(g);
    if (lanes > 4) lanes = 4;
    if (g->amino && lanes < 2) lanes = 2;
    unsigned long long *pos = dPositions;
    const unsigned long long th = totalHits;
    if (g->dev.bwtLength / g->dev.saRatio >= (1ull << 40)) {
      setError("awfmGpuLocate: more than 2^40 suffix-array samples are not supported");
      return AwFmUnsupportedVersionError;
    }
    const bool pow2 = g->dev.saShift != 0xFFFFFFFFu;
    const bool narrow = awfmImageNarrow(g);
    /* two LF steps per block read where the image has its pair blocks (awfm_pair.h); their 32-bit superblock bases are
     * dynamic LDS */
    const bool pair = !g->amino && lanes == 4 && g->dev.pairBlocks;
    const bool superInLds = pair && narrow && awfmPairSuperInLds(g);
    const size_t pairLds = superInLds ? (size_t)g->dev.numPairSuper * (kPairSuperStride * 4u) : 0u;
    DevIndex pairDev = g->dev;
    pairDev.pairSuperInLds = superInLds ? 1u : 0u;
    /* steps after which an uncapped walk is parked for finishKernel to walk on (the hand-over holds 23 bits of steps);
     * $AWFM_GPU_DIAG walk_give_up: a small number, so that the tests reach that path on ordinary texts */
    unsigned giveUp = (1u << kWalkStepBits) - 1u;
    if (const char *env = awfmGpuDiag("walk_give_up")) {
      const long v = atol(env);
      if (v >= 1 && v < (long)giveUp) giveUp = (unsigned)v;
    }
#define AWFM_LOCP(P2, NR)                                                                                                    \
  do {                                                                                                                       \
    const unsigned grid__ = gridFor(th, g, walkKernel<false, 4, P2, NR, true>, walkThreads(true) / 4, pairLds, walkThreads(true)); \
    /* a short hit list: batches of 4 instead of 16 hits per lane group, when the grid has a group for every one */         \
    if (th <= (unsigned long long)grid__ * (walkThreads(true) / 4) * 4ull)                                                   \
      hipLaunchKernelGGL((walkKernel<false, 4, P2, NR, true, 1u>), dim3(grid__), dim3(walkThreads(true)), pairLds, s, pairDev, th, pos, totalOnDevice, stepCap, giveUp); \
    else                                                                                                                     \
      hipLaunchKernelGGL((walkKernel<false, 4, P2, NR, true>), dim3(grid__), dim3(walkThreads(true)), pairLds, s, pairDev, th, pos, totalOnDevice, stepCap, giveUp); \
  } while (0)
#define AWFM_LOC3(AM, GG, P2, NR)                                                                                  \
  hipLaunchKernelGGL((walkKernel<AM, GG, P2, NR>), dim3(gridFor(th, g, walkKernel<AM, GG, P2, NR>, kThreads / GG)), \
                     dim3(kThreads), 0, s, g->dev, th, pos, totalOnDevice, stepCap, giveUp)
#define AWFM_LOC(AM, GG)                                      \
  do {                                                        \
    if (pow2 && narrow) AWFM_LOC3(AM, GG, true, true);        \
    else if (pow2) AWFM_LOC3(AM, GG, true, false);            \
    else if (narrow) AWFM_LOC3(AM, GG, false, true);          \
    else AWFM_LOC3(AM, GG, false, false);                     \
  } while (0)
    if (pair) {
      if (pow2 && narrow) AWFM_LOCP(true, true);
      else if (pow2) AWFM_LOCP(true, false);
      else if (narrow) AWFM_LOCP(false, true);
      else AWFM_LOCP(false, false);
    } else if (g->amino) {
      if (lanes == 4) AWFM_LOC(true, 4);
      else AWFM_LOC(true, 2);
    } else {
      if (lanes == 4) AWFM_LOC(false, 4);
      else if (lanes == 2) AWFM_LOC(false, 2);
      else AWFM_LOC(false, 1);
    }
#undef AWFM_LOC
#undef AWFM_LOC3
#undef AWFM_LOCP
    AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
    /* out-of-place: the final positions go to `out` (page-locked host memory in the pipeline): a smaller grid, so that
     * a kernel paced by the PCIe writes leaves the chip to whatever runs beside it */
    const unsigned finishGrid = out && out != pos ? (unsigned)g->numCUs * 2u : (unsigned)g->numCUs * 8u;
    hipLaunchKernelGGL(finishKernel, dim3(finishGrid), dim3(256), 0, s, g->dev, th, (const unsigned long long *)pos, out ? out : pos, totalOnDevice,
                       stepCap ? 0u : (g->amino ? 2u : 1u), giveUp);
  }
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  return AwFmSuccess;
}
