/*
 * awfm_gpu.hip -- HIP (gfx950 / CDNA4) side of libawfmindex_amd.so.
 *
 * Device image ("re-laid-out windowed BWT", awfm_device.h): blocks of 128 BWT positions -- nucleotide 64 B (three
 * plane words + one 32-bit base count per 32-position slice), amino 128 B (five plane words + six 16-bit base counts
 * per slice) -- with 64-bit base counts kept per superblock (2^32 / 2^16 positions).  The reference layout
 * (ref src/AwFmIndex.h:55-65: 160 / 352 B blocks of 256 positions) straddles two (three to four) 128-B lines per
 * rank; here a rank reads one 64-B granule (one line).
 *
 * Kernels: searchKernel (awfm_search_kernel.h: seed lookup + backward search, G lanes per query),
 * walkKernel/finishKernel (awfm_locate_kernel.h: LF walk to a sampled position, sampled-SA read), and here the
 * hit-offset scan, hit expansion, dense-SA helpers and the layout conversion launches.
 *
 * Semantics restated from the reference (see include/awfm_gpu.h for the map):
 * a query stops at the first invalid range and keeps it; hits are in BWT order.
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

static thread_local std::string tlsError;
void awfmGpuSetError(const char *what) { tlsError = what; }
void awfmGpuSetHipError(const char *what, hipError_t e) { tlsError = std::string(what) + ": " + hipGetErrorString(e); }

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

/* dense device SA construction helpers */
__global__ void iotaKernel(unsigned long long *out, unsigned long long first, unsigned long long count) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = first + i;
}
__global__ void narrowKernel(const unsigned long long *in, unsigned long long count, unsigned *out) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = (unsigned)in[i];
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

/* ------------------------------------------------------------------ host side */

namespace {

std::mutex tableMutex;
struct ImageEntry {
  const AwFmIndex *index;
  int device; /* HIP ordinal the image lives on */
  int lane;   /* 0 = the image itself; n = the n-th extra handle on it (a device named again in $AWFM_GPU_DEVICES) */
  AwFmGpuIndex *image;
};
std::vector<ImageEntry> imageTable;


/* Persistent grid: the kernels stride over the work, so the grid is exactly what is resident
 * (blocksPerCU from the occupancy query for that kernel); a larger grid would run as a second,
 * under-filled round. */
template <class Kernel>
unsigned gridFor(uint64_t groups, const AwFmGpuIndex *g, Kernel kernel, unsigned groupsPerBlock, size_t dynamicLds = 0,
                 int threads = kThreads) {
  int perCU = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kernel, threads, dynamicLds) != hipSuccess || perCU < 1) perCU = 4;
  if (perCU > 8) perCU = 8;
  const uint64_t blocks = (groups + groupsPerBlock - 1) / groupsPerBlock;
  const uint64_t cap = (uint64_t)g->numCUs * (uint64_t)perCU;
  return (unsigned)(blocks < cap ? (blocks ? blocks : 1) : cap);
}

enum AwFmReturnCode ensureWork(AwFmGpuIndex *g, size_t bytes) {
  if (bytes <= g->workBytes) return AwFmSuccess;
  if (g->dWork) (void)hipFree(g->dWork);
  g->dWork = nullptr;
  g->workBytes = 0;
  const size_t want = bytes + bytes / 4 + 4096;
  AWFM_HIP_TRY(hipMalloc(&g->dWork, want), AwFmAllocationFailure);
  g->workBytes = want;
  return AwFmSuccess;
}

inline size_t alignUp(size_t v, size_t a) { return (v + a - 1) / a * a; }

/* the image's full suffix array as a kernel argument */
inline DenseSa denseSaOf(const AwFmGpuIndex *g) {
  DenseSa sa;
  sa.words = (const unsigned *)g->dDenseSa;
  sa.wide = g->denseWide ? 1u : 0u;
  return sa;
}

}  // namespace

namespace {
void fillDevIndex(AwFmGpuIndex *g, const struct AwFmIndex *index, unsigned superShift, unsigned long long sentinelPos) {
  DevIndex &d = g->dev;
  d.blocks = (const uint4 *)g->dBlocks;
  d.super = (const unsigned long long *)g->dSuper;
  d.numSuper = (unsigned)awfmNumSuper(index->bwtLength, index->config.alphabetType == AwFmAlphabetAmino, superShift);
  d.nucSuperShift = superShift;
  d.seed = (const ulonglong2 *)g->dSeed;
  d.sa = (const unsigned long long *)g->dSa;
  d.bwtLength = index->bwtLength;
  d.sentinelPos = sentinelPos;
  d.seedLen = awfmKmerTableLength(index->config.alphabetType, index->config.kmerLengthInSeedTable);
  d.prefixSums = (const unsigned long long *)g->dPrefix;
  d.saRatio = index->config.suffixArrayCompressionRatio;
  d.saShift = 0xFFFFFFFFu;
  if ((d.saRatio & (d.saRatio - 1)) == 0) {
    d.saShift = 0;
    while ((1u << d.saShift) < d.saRatio) d.saShift++;
  }
  d.saWidth = index->suffixArray.valueBitWidth;
  d.seedK = index->config.kmerLengthInSeedTable;
  d.deepSeed = nullptr;
  d.deepK = 0;
  d.deepNarrow = 0;
  d.pairBlocks = nullptr;
  d.pairSuper = nullptr;
  d.pairSuper32 = nullptr;
  d.pairC = nullptr;
  d.numPairSuper = 0;
  d.pairSuperInLds = 0;
}
}  // namespace

namespace {
enum AwFmReturnCode launchLocate(AwFmGpuIndex *g, unsigned long long totalHits, unsigned long long *dPositions,
                                 hipStream_t s, unsigned long long *out = nullptr, const unsigned long long *totalOnDevice = nullptr,
                                 unsigned stepCap = 0u);
/* lanes that cooperate on one query: image setting, else $AWFM_GPU_DIAG kernel=g4|g2|g1, else the default.  A device
 * block has 4 slices, so 4 lanes is the widest group (GROUP8 of the enum maps to it); amino slices are 32 B, 2 lanes
 * per query already hold 64 registers of block data */
int lanesPerQuery(const AwFmGpuIndex *g) {
  int lanes = 4;
  switch (g->kernel) {
    case AWFM_GPU_KERNEL_GROUP8:
    case AWFM_GPU_KERNEL_GROUP4: lanes = 4; break;
    case AWFM_GPU_KERNEL_GROUP2: lanes = 2; break;
    case AWFM_GPU_KERNEL_GROUP1: lanes = 1; break;
    default:
      /* measured on MI355X (scripts/ab_layout.sh): 10^8 random 21-mers against the GRCh38-sized index, general
       * kernel: g4 13.2-13.7 ms, g2 13.0-13.4, g1 13.9 (within the box-to-box spread: the kernel runs at the rate the
       * chip delivers random granules; g4 keeps 8 waves per SIMD without spilling); 5*10^7 amino 10-mers: g4 3.91 ms,
       * g2 3.72 */
      lanes = g->amino ? 2 : 4;
      if (const char *env = awfmGpuDiag("kernel")) { /* lanes per k-mer of the general kernel: g4 | g2 | g1 */
        if (!strcmp(env, "g8") || !strcmp(env, "g4")) lanes = 4;
        else if (!strcmp(env, "g2")) lanes = 2;
        else if (!strcmp(env, "g1")) lanes = 1;
      }
  }
  if (g->amino && lanes < 2) lanes = 2;
  return lanes;
}

/* `dev`: the image view the kernel gets -- g->dev, or a copy with a field changed for this launch only (the tally
 * prices the reference algorithm without the deeper table) so that the shared image is never edited */
template <bool AMINO, int G, bool CSR, bool TALLY, bool NARROW>
void launchSearchKernelN(const AwFmGpuIndex *g, const DevIndex &dev, hipStream_t s, const uint8_t *dChars,
                         const unsigned long long *off, uint32_t fixedLength, unsigned long long nq, ulonglong2 *rng,
                         uint32_t *dCounts, unsigned long long *dTally) {
  const unsigned grid = gridFor(nq, g, searchKernel<AMINO, G, CSR, TALLY, NARROW>, kThreads / G);
  hipLaunchKernelGGL((searchKernel<AMINO, G, CSR, TALLY, NARROW>), dim3(grid), dim3(kThreads), 0, s, dev, dChars, off,
                     fixedLength, nq, rng, dCounts, dTally);
}

/* general search with two characters per block read (searchKernel<..., PAIR>; exact ranges) */
template <bool CSR, bool NARROW>
void launchPairSearchKernel(const AwFmGpuIndex *g, const DevIndex &dev, hipStream_t s, const uint8_t *dChars,
                            const unsigned long long *off, uint32_t fixedLength, unsigned long long nq, ulonglong2 *rng,
                            uint32_t *dCounts) {
  /* the 16 pair bases of every superblock in dynamic LDS, as in the ordered kernel (same box, 10^8 random 21-mers:
   * 11.4-11.7 ms against 12.8-12.9 ms with the bases read from memory; $AWFM_GPU_PAIR_SUPER=lds|global) */
  const bool inLds = NARROW && awfmPairSuperInLds(g);
  const size_t lds = inLds ? (size_t)g->dev.numPairSuper * 64u : 0u;
  DevIndex view = dev;
  view.pairSuperInLds = inLds ? 1u : 0u;
  const unsigned grid = gridFor(nq, g, searchKernel<false, 4, CSR, false, NARROW, false, true>, kThreads / 4, lds);
  hipLaunchKernelGGL((searchKernel<false, 4, CSR, false, NARROW, false, true>), dim3(grid), dim3(kThreads), lds, s, view, dChars,
                     off, fixedLength, nq, rng, dCounts, (unsigned long long *)nullptr);
}

template <bool AMINO, int G, bool CSR, bool TALLY>
void launchSearchKernel(const AwFmGpuIndex *g, const DevIndex &dev, hipStream_t s, const uint8_t *dChars,
                        const unsigned long long *off, uint32_t fixedLength, unsigned long long nq, ulonglong2 *rng,
                        uint32_t *dCounts, unsigned long long *dTally) {
  if (awfmImageNarrow(g))
    launchSearchKernelN<AMINO, G, CSR, TALLY, true>(g, dev, s, dChars, off, fixedLength, nq, rng, dCounts, dTally);
  else
    launchSearchKernelN<AMINO, G, CSR, TALLY, false>(g, dev, s, dChars, off, fixedLength, nq, rng, dCounts, dTally);
}

template <bool TALLY>
void launchSearch(const AwFmGpuIndex *g, const DevIndex &dev, int lanes, hipStream_t s, const uint8_t *dChars,
                  const unsigned long long *off, uint32_t fixedLength, unsigned long long nq, ulonglong2 *rng,
                  uint32_t *dCounts, unsigned long long *dTally) {
#define AWFM_GO(AM, GG)                                                                                            \
  do {                                                                                                             \
    if (off) launchSearchKernel<AM, GG, true, TALLY>(g, dev, s, dChars, off, fixedLength, nq, rng, dCounts, dTally);  \
    else launchSearchKernel<AM, GG, false, TALLY>(g, dev, s, dChars, off, fixedLength, nq, rng, dCounts, dTally);     \
  } while (0)
  if (g->amino) {
    if (lanes == 4) AWFM_GO(true, 4);
    else AWFM_GO(true, 2);
  } else {
    if (lanes == 4) AWFM_GO(false, 4);
    else if (lanes == 2) AWFM_GO(false, 2);
    else AWFM_GO(false, 1);
  }
#undef AWFM_GO
}
}  // namespace

constexpr unsigned kAutoDeepSeedMin = 14, kAutoDeepSeedMax = 16; /* depths of the device-only seed table large nucleotide images get by default */
extern "C" {
static enum AwFmReturnCode applyDeepSeedFromEnv(AwFmGpuIndex *g);
static enum AwFmReturnCode applyPairFromEnv(AwFmGpuIndex *g);
static enum AwFmReturnCode applyDenseSa(AwFmGpuIndex *g, bool enable, bool capped = false);
static enum AwFmReturnCode applyDenseSaFromEnv(AwFmGpuIndex *g);
}

AwFmGpuIndex *awfmGpuIndexAdopt(const struct AwFmIndex *index, int device, void *dBlocks, void *dSuper, unsigned superShift,
                                void *dSeed, void *dSa, void *dPrefix, unsigned long long sentinelPos, uint64_t deviceBytes) {
  AwFmGpuIndex *g = new AwFmGpuIndex();
  g->device = device;
  g->amino = index->config.alphabetType == AwFmAlphabetAmino;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
    g->numCUs = prop.multiProcessorCount;
  g->numBlocks = awfmDeviceBlocks(index->bwtLength);
  g->dBlocks = dBlocks;
  g->dSuper = dSuper;
  g->dSeed = dSeed;
  g->dSa = dSa;
  g->dPrefix = dPrefix;
  g->deviceBytes = deviceBytes;
  fillDevIndex(g, index, superShift, sentinelPos);
  if (const char *env = getenv("AWFM_GPU_FORCE_WIDE")) g->forceWide = atoi(env) != 0;
  (void)applyPairFromEnv(g);     /* first: the deeper table's next-step bits are computed through the pair image */
  (void)applyDeepSeedFromEnv(g); /* optional accelerator: on failure the image simply has no deeper table */
  (void)applyDenseSaFromEnv(g);  /* the same: without it a locate walks */
  return g;
}

void awfmGpuIndexRegister(const struct AwFmIndex *index, AwFmGpuIndex *g) {
  std::lock_guard<std::mutex> lock(tableMutex);
  imageTable.push_back({index, g->device, 0, g});
}

bool awfmGpuRelayout(const void *dRefBlocks, uint64_t bwtLength, bool amino, unsigned superShift, void *dBlocks,
                     void *dSuper, unsigned long long *sentinelPosOut) {
  const uint64_t numRef = awfmNumBlocks(bwtLength);
  const unsigned numSuper = (unsigned)awfmNumSuper(bwtLength, amino, superShift);
  unsigned long long *dSentinel = nullptr;
  hipError_t e = hipMalloc((void **)&dSentinel, 8);
  if (e == hipSuccess) e = hipMemset(dSentinel, 0, 8);
  if (e == hipSuccess) {
    const unsigned words = numSuper * (amino ? kAminoSuperStride : 4u);
    hipLaunchKernelGGL(gatherSuperKernel, dim3((words + 255) / 256), dim3(256), 0, 0, (const unsigned long long *)dRefBlocks,
                       (unsigned long long)numRef, amino ? 1 : 0, superShift, numSuper, (unsigned long long *)dSuper);
    const uint64_t threads = numRef * 2 * kSlices;
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (amino)
      hipLaunchKernelGGL(relayoutAminoKernel, dim3(grid), dim3(256), 0, 0, (const unsigned long long *)dRefBlocks,
                         (unsigned long long)numRef, (unsigned long long)bwtLength, (const unsigned long long *)dSuper,
                         (uint4 *)dBlocks, dSentinel);
    else
      hipLaunchKernelGGL(relayoutNucKernel, dim3(grid), dim3(256), 0, 0, (const unsigned long long *)dRefBlocks,
                         (unsigned long long)numRef, (unsigned long long)bwtLength, superShift,
                         (const unsigned long long *)dSuper, (uint4 *)dBlocks, dSentinel);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(sentinelPosOut, dSentinel, 8, hipMemcpyDeviceToHost);
  if (dSentinel) (void)hipFree(dSentinel);
  if (e != hipSuccess) {
    setError("awfmGpuRelayout", e);
    return false;
  }
  return true;
}

extern "C" {

int awfmGpuDeviceCount(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *awfmGpuLastError(void) { return tlsError.c_str(); }

enum AwFmReturnCode awfmGpuIndexCreate(const struct AwFmIndex *index, int device, AwFmGpuIndex **out) {
  if (!index || !out) {
    setError("awfmGpuIndexCreate: null argument");
    return AwFmNullPtrError;
  }
  *out = nullptr;
  if (awfmGpuDeviceCount() <= 0) {
    setError("awfmGpuIndexCreate: no HIP device available (this library has no CPU search path)");
    return AwFmGeneralFailure;
  }
  if (device < 0) {
    const char *env = getenv("AWFM_GPU_DEVICE");
    if (env && *env) {
      device = atoi(env);
    } else if (hipGetDevice(&device) != hipSuccess) {
      device = 0;
    }
  }
  DeviceGuard guard(device);
  if (!guard.ok) {
    setError("awfmGpuIndexCreate: hipSetDevice failed");
    return AwFmGeneralFailure;
  }
  const bool amino = index->config.alphabetType == AwFmAlphabetAmino;
  const unsigned superShift = awfmSuperShift(amino, index->bwtLength);
  if (!amino && awfmNumSuper(index->bwtLength, false, superShift) > kMaxNucSuper) {
    setError("awfmGpuIndexCreate: nucleotide device images hold at most 64 superblocks (2^38 positions)");
    return AwFmUnsupportedVersionError;
  }
  if (index->config.suffixArrayCompressionRatio == 0) {
    setError("awfmGpuIndexCreate: suffixArrayCompressionRatio must be >= 1");
    return AwFmGeneralFailure;
  }

  AwFmGpuIndex *g = new AwFmGpuIndex();
  g->device = device;
  g->amino = amino;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
    g->numCUs = prop.multiProcessorCount;
  g->numBlocks = awfmDeviceBlocks(index->bwtLength);
  const size_t refBytes = awfmNumBlocks(index->bwtLength) * awfmBlockBytes(index->config.alphabetType);
  const size_t devBlockBytes = g->numBlocks * awfmDeviceBlockBytes(amino);
  const size_t superBytes = awfmSuperBytes(index->bwtLength, amino, superShift);
  const uint64_t seedLen = awfmKmerTableLength(index->config.alphabetType, index->config.kmerLengthInSeedTable);
  const size_t seedBytes = seedLen * sizeof(struct AwFmSearchRange);
  const size_t saBytes = index->suffixArray.compressedByteLength;
  const size_t saAlloc = alignUp(saBytes, 16) + 256; /* the locate kernel reads a 128-byte window at a sample */

  auto fail = [&](enum AwFmReturnCode rc) {
    awfmGpuIndexDestroy(g);
    return rc;
  };
  void *dRef = nullptr;
#define TRY_OR_FAIL(call, rc)                 \
  do {                                        \
    hipError_t e__ = (call);                  \
    if (e__ != hipSuccess) {                  \
      setError(#call, e__);                   \
      if (dRef) (void)hipFree(dRef);          \
      return fail(rc);                        \
    }                                         \
  } while (0)

  TRY_OR_FAIL(hipMalloc(&g->dBlocks, devBlockBytes), AwFmAllocationFailure);
  TRY_OR_FAIL(hipMalloc(&g->dSuper, superBytes), AwFmAllocationFailure);
  TRY_OR_FAIL(hipMalloc(&g->dSeed, seedBytes ? seedBytes : 16), AwFmAllocationFailure);
  TRY_OR_FAIL(hipMalloc(&g->dSa, saAlloc), AwFmAllocationFailure);
  TRY_OR_FAIL(hipMalloc(&dRef, refBytes), AwFmAllocationFailure);
  g->deviceBytes = devBlockBytes + superBytes + seedBytes + saAlloc;

  TRY_OR_FAIL(hipMemcpy(dRef, index->bwtBlockList.asNucleotide, refBytes, hipMemcpyHostToDevice), AwFmGeneralFailure);
  unsigned long long sentinelPos = 0;
  if (!awfmGpuRelayout(dRef, index->bwtLength, amino, superShift, g->dBlocks, g->dSuper, &sentinelPos)) {
    (void)hipFree(dRef);
    return fail(AwFmGeneralFailure);
  }
  (void)hipFree(dRef);
  dRef = nullptr;

  TRY_OR_FAIL(hipMemcpy(g->dSeed, index->kmerSeedTable, seedBytes, hipMemcpyHostToDevice), AwFmGeneralFailure);
  {
    unsigned long long prefix[24] = {0};
    memcpy(prefix, index->prefixSums, awfmPrefixSumsLength(index->config.alphabetType) * sizeof(uint64_t));
    TRY_OR_FAIL(hipMalloc(&g->dPrefix, sizeof prefix), AwFmAllocationFailure);
    TRY_OR_FAIL(hipMemcpy(g->dPrefix, prefix, sizeof prefix, hipMemcpyHostToDevice), AwFmGeneralFailure);
  }

  /* sampled SA: from memory, or staged from the index file (keepSuffixArrayInMemory == false) */
  TRY_OR_FAIL(hipMemset(g->dSa, 0, saAlloc), AwFmGeneralFailure);
  if (index->suffixArray.values) {
    TRY_OR_FAIL(hipMemcpy(g->dSa, index->suffixArray.values, saBytes, hipMemcpyHostToDevice), AwFmGeneralFailure);
  } else {
    uint8_t *staged = awfmReadPackedSaFromFile(index);
    if (!staged) {
      setError("awfmGpuIndexCreate: index has no in-memory suffix array and it could not be read from its file");
      return fail(AwFmFileReadFail);
    }
    hipError_t e = hipMemcpy(g->dSa, staged, saBytes, hipMemcpyHostToDevice);
    free(staged);
    TRY_OR_FAIL(e, AwFmGeneralFailure);
  }
#undef TRY_OR_FAIL

  fillDevIndex(g, index, superShift, sentinelPos);
  if (const char *env = getenv("AWFM_GPU_FORCE_WIDE")) g->forceWide = atoi(env) != 0;
  (void)applyPairFromEnv(g); /* without it (no memory left) searches simply take one step per read */
  if (applyDeepSeedFromEnv(g) != AwFmSuccess) return fail(AwFmGeneralFailure);
  (void)applyDenseSaFromEnv(g); /* optional accelerator: without it (no memory left) a locate walks */
  *out = g;
  return AwFmSuccess;
}

void awfmGpuIndexDestroy(AwFmGpuIndex *g) {
  if (!g) return;
  {
    DeviceGuard guard(g->device);
    awfmGpuStreamStateFree(g);
    if (!g->shares) { /* a lane owns only its staging */
      if (g->dBlocks) (void)hipFree(g->dBlocks);
      if (g->dSuper) (void)hipFree(g->dSuper);
      if (g->dSeed) (void)hipFree(g->dSeed);
      if (g->dSa) (void)hipFree(g->dSa);
      if (g->dPrefix) (void)hipFree(g->dPrefix);
      if (g->dDeepSeed) (void)hipFree(g->dDeepSeed);
      if (g->dDeepBig) (void)hipFree(g->dDeepBig);
      if (g->dDenseSa) (void)hipFree(g->dDenseSa);
      if (g->dLengthTable) (void)hipFree(g->dLengthTable);
      if (g->dLengthBig) (void)hipFree(g->dLengthBig);
      void *pairOwned[] = {g->dPairBlocks, g->dPairSuper, g->dPairSuper32, g->dPairC};
      for (void *p : pairOwned)
        if (p) (void)hipFree(p);
    }
    if (g->dWork) (void)hipFree(g->dWork);
    if (g->dHits) (void)hipFree(g->dHits);
    for (auto &slot : g->orderSlot) {
      if (slot.mem) (void)hipFree(slot.mem);
      if (slot.gate.done) (void)hipEventDestroy(slot.gate.done);
    }
    if (g->dSparse) (void)hipFree(g->dSparse);
    if (g->sparseGate.done) (void)hipEventDestroy(g->sparseGate.done);
    for (auto &entry : g->orderLog)
      for (int i = 0; i < 4; i++)
        if (entry.ev[i]) (void)hipEventDestroy(entry.ev[i]);
    for (int i = 0; i < 2; i++)
      if (g->windowEvent[i]) (void)hipEventDestroy(g->windowEvent[i]);
    for (int i = 0; i < 4; i++)
      if (g->pinned[i]) (void)hipHostFree(g->pinned[i]);
    if (g->predict.verdictHost) (void)hipHostFree(g->predict.verdictHost);
  }
  delete g;
}

/* device ordinals the AoS entry points shard over: $AWFM_GPU_DEVICES = "all" or a comma list (a device named
 * again gets a lane on its image); unset = the default device (-1) with three lanes, so that one chunk of a
 * list is packed / scattered on the host while others are on the PCIe bus or in the kernels (awfm_batch.c) */
static int aosDevices(int *devs, int maxOut) {
  int n = 0;
  const char *env = getenv("AWFM_GPU_DEVICES");
  if (env && !strcmp(env, "all")) {
    const int count = awfmGpuDeviceCount();
    for (int d = 0; d < count && n < maxOut; d++) devs[n++] = d;
  } else if (env && *env) {
    for (const char *c = env; *c && n < maxOut;) {
      devs[n++] = atoi(c);
      while (*c && *c != ',') c++;
      if (*c == ',') c++;
    }
  }
  if (n == 0) { /* three lanes on the default device: one packs or scatters while two are in their device stage */
    devs[n++] = -1;
    for (int lane = 1; lane < 3 && n < maxOut; lane++) devs[n++] = -1;
  }
  return n;
}

/* the lanes of a primary image (call with tableMutex NOT held) */
static std::vector<AwFmGpuIndex *> lanesOf(const AwFmGpuIndex *primary) {
  std::vector<AwFmGpuIndex *> lanes;
  std::lock_guard<std::mutex> lock(tableMutex);
  for (auto &e : imageTable)
    if (e.image->shares == primary) lanes.push_back(e.image);
  return lanes;
}

static AwFmGpuIndex *makeLane(AwFmGpuIndex *primary) {
  AwFmGpuIndex *g = new AwFmGpuIndex();
  g->shares = primary;
  g->device = primary->device;
  g->amino = primary->amino;
  g->dev = primary->dev;
  g->dBlocks = primary->dBlocks;
  g->dSuper = primary->dSuper;
  g->dSeed = primary->dSeed;
  g->dSa = primary->dSa;
  g->dPrefix = primary->dPrefix;
  g->dDeepSeed = primary->dDeepSeed;
  g->dDenseSa = primary->dDenseSa;
  g->denseWide = primary->denseWide;
  g->numBlocks = primary->numBlocks;
  g->kernel = primary->kernel;
  g->forceWide = primary->forceWide;
  g->numCUs = primary->numCUs;
  return g;
}

int awfmGpuIndexAcquireAll(const struct AwFmIndex *index, AwFmGpuIndex **out, int maxOut) {
  int devs[64];
  const int numDevs = aosDevices(devs, 64);
  /* -1 = the default device: $AWFM_GPU_DEVICE, else the calling thread's current device.  Entries are keyed by
   * the resolved ordinal, so a list that changes between calls never hands out another device's image. */
  int fallback = 0;
  if (const char *env = getenv("AWFM_GPU_DEVICE"); env && *env) fallback = atoi(env);
  else if (hipGetDevice(&fallback) != hipSuccess) fallback = 0;
  for (int i = 0; i < numDevs; i++)
    if (devs[i] < 0) devs[i] = fallback;
  std::lock_guard<std::mutex> lock(tableMutex);
  auto find = [&](int device, int lane) -> AwFmGpuIndex * {
    for (auto &e : imageTable)
      if (e.index == index && e.device == device && e.lane == lane) return e.image;
    return nullptr;
  };
  int n = 0;
  for (int slot = 0; slot < numDevs && n < maxOut; slot++) {
    int lane = 0; /* how often this device was named before */
    for (int earlier = 0; earlier < slot; earlier++) lane += devs[earlier] == devs[slot];
    AwFmGpuIndex *g = find(devs[slot], lane);
    if (!g) {
      if (lane > 0) { /* a device named again gets a lane on the image it already has */
        AwFmGpuIndex *primary = find(devs[slot], 0);
        if (!primary) return n;
        g = makeLane(primary);
      } else if (awfmGpuIndexCreate(index, devs[slot], &g) != AwFmSuccess) {
        return n;
      }
      imageTable.push_back({index, devs[slot], lane, g});
    }
    out[n++] = g;
  }
  return n;
}

AwFmGpuIndex *awfmGpuIndexAcquire(const struct AwFmIndex *index) {
  AwFmGpuIndex *g = nullptr;
  return awfmGpuIndexAcquireAll(index, &g, 1) == 1 ? g : nullptr;
}

void awfmGpuIndexRelease(const struct AwFmIndex *index) {
  std::vector<AwFmGpuIndex *> doomed;
  {
    std::lock_guard<std::mutex> lock(tableMutex);
    for (size_t i = 0; i < imageTable.size();) {
      if (imageTable[i].index == index) {
        doomed.push_back(imageTable[i].image);
        imageTable.erase(imageTable.begin() + (long)i);
      } else {
        i++;
      }
    }
  }
  for (AwFmGpuIndex *g : doomed)
    if (g->shares) awfmGpuIndexDestroy(g); /* lanes first: they point into their primary */
  for (AwFmGpuIndex *g : doomed)
    if (!g->shares) awfmGpuIndexDestroy(g);
}

void *awfmGpuPinnedBuffer(AwFmGpuIndex *g, int slot, uint64_t bytes) {
  if (!g || slot < 0 || slot > 3) return nullptr;
  if (bytes <= g->pinnedBytes[slot]) return g->pinned[slot];
  DeviceGuard guard(g->device);
  if (g->pinned[slot]) (void)hipHostFree(g->pinned[slot]);
  g->pinned[slot] = nullptr;
  g->pinnedBytes[slot] = 0;
  const size_t want = bytes + bytes / 4 + 4096;
  if (hipHostMalloc(&g->pinned[slot], want, hipHostMallocDefault) != hipSuccess) {
    setError("awfmGpuPinnedBuffer: hipHostMalloc failed");
    g->pinned[slot] = nullptr;
    return nullptr;
  }
  g->pinnedBytes[slot] = want;
  return g->pinned[slot];
}
void awfmGpuAosLock(AwFmGpuIndex *g) {
  if (g) g->aosMutex.lock();
}
void awfmGpuAosUnlock(AwFmGpuIndex *g) {
  if (g) g->aosMutex.unlock();
}

uint64_t awfmGpuIndexDeviceBytes(const AwFmGpuIndex *g) {
  return g ? g->deviceBytes + g->deepSeedBytes + g->denseSaBytes + g->pairBytes + g->lengthTableBytes : 0;
}

namespace {
/* holds the work and AoS locks of every lane of a primary image for the lifetime of the object */
struct LaneLocks {
  std::vector<AwFmGpuIndex *> lanes;
  explicit LaneLocks(const AwFmGpuIndex *primary) : lanes(lanesOf(primary)) {
    for (AwFmGpuIndex *lane : lanes) {
      lane->aosMutex.lock();
      lane->workMutex.lock();
    }
  }
  ~LaneLocks() {
    for (AwFmGpuIndex *lane : lanes) {
      lane->workMutex.unlock();
      lane->aosMutex.unlock();
    }
  }
};
}  // namespace

/* replaces the deeper table of a primary image and of the given lanes; the caller holds whatever locks the image
 * needs (none for an image nobody else has a pointer to yet) */
static enum AwFmReturnCode applyDeepSeed(AwFmGpuIndex *g, unsigned deepK, const std::vector<AwFmGpuIndex *> &laneList);

enum AwFmReturnCode awfmGpuIndexSetDeepSeed(AwFmGpuIndex *g, unsigned deepK) {
  if (!g) {
    setError("awfmGpuIndexSetDeepSeed: null image");
    return AwFmNullPtrError;
  }
  if (g->shares) {
    setError("awfmGpuIndexSetDeepSeed: set it on the primary image, not on a lane");
    return AwFmIllegalPositionError;
  }
  DeviceGuard guard(g->device);
  LaneLocks lanes(g); /* nobody searches through a lane while the table is replaced */
  std::lock_guard<std::mutex> lock(g->workMutex);
  return applyDeepSeed(g, deepK, lanes.lanes);
}

static enum AwFmReturnCode applyDeepSeed(AwFmGpuIndex *g, unsigned deepK, const std::vector<AwFmGpuIndex *> &laneList) {
  (void)hipDeviceSynchronize();
  if (g->dDeepSeed) (void)hipFree(g->dDeepSeed);
  if (g->dDeepBig) (void)hipFree(g->dDeepBig);
  g->dDeepSeed = nullptr;
  g->dDeepBig = nullptr;
  g->deepSeedBytes = 0;
  g->dev.deepSeed = nullptr;
  g->dev.deepK = 0;
  g->dev.deepNarrow = 0;
  g->dev.deepNext = 0;
  g->dev.numDeepBig = 0;
  g->dev.deepBigBySp = nullptr;
  { /* the tables of the shorter lengths go with the deeper table they complete; the next mixed-length batch builds them again */
    std::lock_guard<std::mutex> lock(g->lengthMutex);
    if (g->dLengthTable) (void)hipFree(g->dLengthTable);
    if (g->dLengthBig) (void)hipFree(g->dLengthBig);
    g->dLengthTable = nullptr;
    g->dLengthBig = nullptr;
    g->lengthDepths = 0;
    g->lengthTableBytes = 0;
    g->lengthTried = false;
  }
  enum AwFmReturnCode rc = AwFmSuccess;
  g->deepSeedBuildSeconds = 0.0;
  g->deepSeedTransientBytes = 0;
  if (deepK != 0) {
    void *table = nullptr;
    uint64_t bytes = 0, peak = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    unsigned format = 0;
    void *big = nullptr;
    if (awfmGpuBuildDeepSeedTable(g, deepK, &table, &bytes, &peak, &g->deepSeedAllocSeconds, &format, &big)) {
      unsigned numBig = 0;
      /* the next-step bits: images with pair blocks (format 1: the long lengths move to `big` with them) */
      const int next = awfmGpuDeepSeedAddNext(g, table, deepK, format, &big, &numBig);
      (void)hipDeviceSynchronize();
      clock_gettime(CLOCK_MONOTONIC, &t1);
      g->deepSeedBuildSeconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
      g->deepSeedTransientBytes = peak > bytes ? peak - bytes : 0;
      if (next < 0) {
        (void)hipFree(table);
        if (big) (void)hipFree(big);
        rc = AwFmGeneralFailure;
      } else {
        g->dDeepSeed = table;
        g->dDeepBig = big;
        const uint64_t bigBytes = !big ? 0u
                                  : format == 2u ? ((g->dev.bwtLength >> kDeepWideBigShift) + 2u) * 8u
                                                 : ((g->dev.bwtLength >> (g->amino ? kAminoDeepBigShift : kDeepBigShift)) + 5u) * 4u;
        g->deepSeedBytes = bytes + bigBytes;
        g->dev.deepSeed = (const ulonglong2 *)table;
        g->dev.deepK = deepK;
        g->dev.deepNarrow = format;
        g->dev.deepNext = next > 0 ? 1u : 0u;
        g->dev.numDeepBig = numBig;
        g->dev.deepBigBySp = (const unsigned *)big;
        if (getenv("AWFM_VERBOSE"))
          fprintf(stderr, "[awfm deeper table] depth %u, entry format %u: %.2f GB in %.2f s; next-step bits %s; %u entries with long ranges\n",
                  deepK, format, (double)bytes * 1e-9, g->deepSeedBuildSeconds, next > 0 ? "yes" : "no", numBig);
      }
    } else {
      rc = AwFmGeneralFailure;
    }
  }
  for (AwFmGpuIndex *lane : laneList) {
    lane->dDeepSeed = g->dDeepSeed;
    lane->dev.deepSeed = g->dev.deepSeed;
    lane->dev.deepNarrow = g->dev.deepNarrow;
    lane->dev.deepK = g->dev.deepK;
    lane->dev.deepNext = g->dev.deepNext;
    lane->dev.numDeepBig = g->dev.numDeepBig;
    lane->dev.deepBigBySp = g->dev.deepBigBySp;
  }
  return rc;
}

/* $AWFM_GPU_DEEP_SEED_K on an image that was just created or adopted: nobody else holds it and it has no lanes,
 * so no lock is taken -- awfmGpuIndexAcquireAll creates images while it holds the table lock, and the public
 * setter would ask for that lock again through lanesOf() */
static enum AwFmReturnCode applyDeepSeedFromEnv(AwFmGpuIndex *g) {
  int deepK = 0;
  if (const char *env = getenv(g->amino ? "AWFM_GPU_AMINO_DEEP_SEED_K" : "AWFM_GPU_DEEP_SEED_K")) {
    deepK = atoi(env); /* 0: none */
  } else if (g->amino) {
    /* Automatic, amino: an image of >= 2^26 positions whose own table is shallower gets the deepest table of up to 7
     * characters with at most 8 entries per text position, when three times its size is free on the device: 20^7 x 8 B =
     * 10.2 GB for a Swiss-Prot-sized text (2 * 10^8 residues), where 85 % of random 10-mers end at their entry (no such
     * 7-mer) and the rest start two steps further on.  Exact: an entry is what the stepping holds after those steps. */
    size_t freeBytes = 0, totalBytes = 0;
    DeviceGuard guard(g->device);
    if (g->dev.bwtLength >= (1ull << 26) && g->dev.bwtLength < (1ull << 32) && g->dev.seedK >= 2 && hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess) {
      unsigned long long entries = 1;
      for (unsigned k = 1; k <= 7u; k++) {
        entries *= 20ull;
        if (k > g->dev.seedK && entries <= 8ull * g->dev.bwtLength && freeBytes / 3u >= entries * 8ull) deepK = (int)k;
        else if (k > g->dev.seedK && entries <= 8ull * g->dev.bwtLength && k > (unsigned)deepK) g->accelNotes += "deeper table: depth " + std::to_string(k) + " not built (less than 3 x its size free); ";
      }
    } else {
      (void)hipGetLastError();
    }
  } else if (g->dev.bwtLength >= (1ull << 28) && g->dev.seedK >= 8 && g->dev.seedK < kAutoDeepSeedMin) {
    /* Automatic: an image far beyond the L2s gets the deepest table of 14..16 characters that has no more than two
     * entries per text position, when the device has room to spare (8 B -- 16 B from 2^32 positions -- x 4^K: 2.1 GB
     * at 14, 34 GB at 16; its construction holds the level below beside it; asked for: three times the table).  Every
     * level of the table replaces a dependent block read of EVERY k-mer by a wider spread of the one table read: 10^8
     * random 21-mers against a 3.1 Gbp image, seed-order search kernel 3.44 ms at 14, 3.21 at 15, 2.87 at 16 (the
     * index's own k = 12 table: 4.6); planted 21-mers 6.08 -> 5.13 ms.  Results are bit-identical (the table holds
     * what the stepping would compute, stop-at-first-invalid rule included). */
    size_t freeBytes = 0, totalBytes = 0;
    DeviceGuard guard(g->device);
    if (hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess) {
      const uint64_t entryBytes = g->dev.bwtLength < (1ull << kDeepWideMaxBits) ? 8u : 16u;
      for (unsigned k = kAutoDeepSeedMax; k >= kAutoDeepSeedMin && deepK == 0; k--)
        if ((1ull << (2u * k)) <= 2ull * g->dev.bwtLength && freeBytes / 3u >= (entryBytes << (2u * k))) deepK = (int)k;
      if (deepK == 0 && freeBytes / 4u >= (16ull << (2u * kAutoDeepSeedMin))) deepK = (int)kAutoDeepSeedMin;
      unsigned wanted = 0; /* the depth the image's size asks for */
      for (unsigned k = kAutoDeepSeedMax; k >= kAutoDeepSeedMin && wanted == 0; k--)
        if ((1ull << (2u * k)) <= 2ull * g->dev.bwtLength) wanted = k;
      if ((unsigned)deepK < wanted)
        g->accelNotes += "deeper table: depth " + std::to_string(wanted) + " not built (less than 3 x its size free)" +
                         (deepK ? ", depth " + std::to_string(deepK) + " instead; " : "; ");
    } else {
      (void)hipGetLastError();
    }
  }
  if (deepK <= 0 || (unsigned)deepK <= g->dev.seedK) return AwFmSuccess; /* nothing deeper than the index's own table */
  DeviceGuard guard(g->device);
  return applyDeepSeed(g, (unsigned)deepK, {});
}
/* Pair image (awfm_pair.h) of a nucleotide image that was just created or adopted (nobody else holds it, no lanes, so
 * no lock): built unless $AWFM_GPU_PAIR=0.  It doubles the block bytes of the image (128 B per 128 positions beside
 * the 64 B of the one-letter blocks) and halves the dependent block reads of hits-only searches and of the LF walk. */
static enum AwFmReturnCode applyPairFromEnv(AwFmGpuIndex *g) {
  if (g->amino) return AwFmSuccess;
  if (const char *env = getenv("AWFM_GPU_PAIR"))
    if (atoi(env) == 0) return AwFmSuccess;
  DeviceGuard guard(g->device);
  const enum AwFmReturnCode rc = awfmGpuApplyPairImage(g, true);
  if (rc != AwFmSuccess) {
    (void)awfmGpuApplyPairImage(g, false);
    g->accelNotes += "pair image: not built (no device memory for 1 byte per position); ";
  }
  return rc;
}

enum AwFmReturnCode awfmGpuIndexSetPairImage(AwFmGpuIndex *g, int enable) {
  if (!g) {
    setError("awfmGpuIndexSetPairImage: null image");
    return AwFmNullPtrError;
  }
  if (g->shares) {
    setError("awfmGpuIndexSetPairImage: set it on the primary image, not on a lane");
    return AwFmIllegalPositionError;
  }
  DeviceGuard guard(g->device);
  LaneLocks lanes(g); /* nobody searches through a lane while the image changes */
  std::lock_guard<std::mutex> lock(g->workMutex);
  const enum AwFmReturnCode rc = awfmGpuApplyPairImage(g, enable != 0);
  if (rc != AwFmSuccess) (void)awfmGpuApplyPairImage(g, false);
  for (AwFmGpuIndex *lane : lanes.lanes) {
    lane->dev.pairBlocks = g->dev.pairBlocks;
    lane->dev.pairSuper = g->dev.pairSuper;
    lane->dev.pairSuper32 = g->dev.pairSuper32;
    lane->dev.pairC = g->dev.pairC;
    lane->dev.numPairSuper = g->dev.numPairSuper;
  }
  return rc;
}
int awfmGpuIndexHasPairImage(const AwFmGpuIndex *g) { return g && g->dev.pairBlocks ? 1 : 0; }
unsigned awfmGpuIndexDeepSeedK(const AwFmGpuIndex *g) { return g ? g->dev.deepK : 0u; }
double awfmGpuIndexDeepSeedAllocSeconds(const AwFmGpuIndex *g) { return g ? (g->shares ? g->shares : g)->deepSeedAllocSeconds : 0.0; }
double awfmGpuIndexDeepSeedBuildSeconds(const AwFmGpuIndex *g) { return g ? (g->shares ? g->shares : g)->deepSeedBuildSeconds : 0.0; }
uint64_t awfmGpuIndexDeepSeedTransientBytes(const AwFmGpuIndex *g) { return g ? (g->shares ? g->shares : g)->deepSeedTransientBytes : 0; }

int awfmGpuIndexDevice(const AwFmGpuIndex *g) { return g ? g->device : -1; }
void awfmGpuIndexSetKernel(AwFmGpuIndex *g, enum AwFmGpuKernel kernel) {
  if (g) g->kernel = kernel;
}
int awfmGpuIndexIsWide(const AwFmGpuIndex *g) { return g && !awfmImageNarrow(g) ? 1 : 0; }
void awfmGpuIndexSetWide(AwFmGpuIndex *g, int wide) {
  if (!g) return;
  g->forceWide = wide != 0;
  if (!g->shares)
    for (AwFmGpuIndex *lane : lanesOf(g)) lane->forceWide = g->forceWide;
}

static enum AwFmReturnCode searchGeneral(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                         uint32_t fixedLength, uint64_t numQueries, struct AwFmSearchRange *dRanges,
                                         uint32_t *dCounts, void *stream);

enum AwFmReturnCode awfmGpuSearch(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                  uint32_t fixedLength, uint64_t numQueries, struct AwFmSearchRange *dRanges,
                                  uint32_t *dCounts, void *stream) {
  return searchGeneral(g, dChars, dOffsets, fixedLength, numQueries, dRanges, dCounts, stream);
}

/* the general kernel: exact ranges (the reference's final range for k-mers without hits too).  Nucleotide images with
 * pair blocks take two characters per block read (exact as well, awfm_pair.h). */
static enum AwFmReturnCode searchGeneral(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                         uint32_t fixedLength, uint64_t numQueries, struct AwFmSearchRange *dRanges,
                                         uint32_t *dCounts, void *stream) {
  if (!g) {
    setError("awfmGpuSearch: null image");
    return AwFmNullPtrError;
  }
  if (numQueries == 0) return AwFmSuccess;
  if (!dChars || (!dOffsets && fixedLength == 0)) {
    setError("awfmGpuSearch: queries need dChars and either dOffsets or fixedLength");
    return AwFmNullPtrError;
  }
  DeviceGuard guard(g->device);
  hipStream_t s = (hipStream_t)stream;
  const int lanes = lanesPerQuery(g);
  if (!g->amino && lanes == 4) {
    /* large batches on an image with its device-only tables: the exact range of every k-mer from one table entry and the
     * few steps behind it (awfm_exact_lookup_kernel.h); no scratch memory for it: the general kernel needs none */
    const int did = awfmGpuExactLookupSearch(g, s, dChars, (const unsigned long long *)dOffsets, fixedLength, numQueries, (ulonglong2 *)dRanges, dCounts);
    g->lastSearchExact = did > 0 ? 1 : 0;
    if (did > 0) return AwFmSuccess;
    if (did < 0 && did != -(int)AwFmAllocationFailure) return (enum AwFmReturnCode)(-did);
  }
  if (!g->amino && lanes == 4 && g->dev.pairBlocks) {
    const unsigned long long *off = (const unsigned long long *)dOffsets;
    const bool narrow = awfmImageNarrow(g);
#define AWFM_PAIR_GO(CSRV, NR) \
  launchPairSearchKernel<CSRV, NR>(g, g->dev, s, dChars, off, fixedLength, numQueries, (ulonglong2 *)dRanges, dCounts)
    if (off) narrow ? AWFM_PAIR_GO(true, true) : AWFM_PAIR_GO(true, false);
    else narrow ? AWFM_PAIR_GO(false, true) : AWFM_PAIR_GO(false, false);
#undef AWFM_PAIR_GO
  } else {
    launchSearch<false>(g, g->dev, lanes, s, dChars, (const unsigned long long *)dOffsets, fixedLength, numQueries,
                        (ulonglong2 *)dRanges, dCounts, nullptr);
  }
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  return AwFmSuccess;
}

/* Hits-only search (include/awfm_gpu.h): ordered path when it applies, else the general kernel, whose exact
 * empty ranges satisfy the contract as well */
static enum AwFmReturnCode searchHits(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets, uint32_t fixedLength,
                                      uint64_t numQueries, struct AwFmSearchRange *dRanges, uint32_t *dCounts, void *stream,
                                      bool rangesOfHitsOnly);

enum AwFmReturnCode awfmGpuSearchHits(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                      uint32_t fixedLength, uint64_t numQueries, struct AwFmSearchRange *dRanges,
                                      uint32_t *dCounts, void *stream) {
  return searchHits(g, dChars, dOffsets, fixedLength, numQueries, dRanges, dCounts, stream, false);
}

/* include/awfm_gpu.h: counts for every k-mer, ranges for the k-mers with hits only */
enum AwFmReturnCode awfmGpuSearchHitsSparse(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                            uint32_t fixedLength, uint64_t numQueries, struct AwFmSearchRange *dRanges,
                                            uint32_t *dCounts, void *stream) {
  if (!dCounts && numQueries) {
    setError("awfmGpuSearchHitsSparse: the counts are what says which ranges were written: dCounts must not be NULL");
    return AwFmNullPtrError;
  }
  return searchHits(g, dChars, dOffsets, fixedLength, numQueries, dRanges, dCounts, stream, true);
}

static enum AwFmReturnCode searchHits(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets, uint32_t fixedLength,
                                      uint64_t numQueries, struct AwFmSearchRange *dRanges, uint32_t *dCounts, void *stream,
                                      bool rangesOfHitsOnly) {
  if (!g) {
    setError("awfmGpuSearchHits: null image");
    return AwFmNullPtrError;
  }
  if (numQueries == 0) return AwFmSuccess;
  if (!dChars || (!dOffsets && fixedLength == 0)) {
    setError("awfmGpuSearchHits: queries need dChars and either dOffsets or fixedLength");
    return AwFmNullPtrError;
  }
  if (g->kernel == AWFM_GPU_KERNEL_AUTO || g->kernel == AWFM_GPU_KERNEL_GROUP4) {
    DeviceGuard guard(g->device);
    const int ordered = awfmGpuOrderedSearch(g, (hipStream_t)stream, dChars, (const unsigned long long *)dOffsets, fixedLength,
                                             numQueries, (ulonglong2 *)dRanges, dCounts, false, rangesOfHitsOnly);
    /* (no memory for the seed-order scratch: the general kernel needs none and gives the same hits) */
    if (ordered < 0 && ordered != -(int)AwFmAllocationFailure) return (enum AwFmReturnCode)(-ordered);
    if (ordered > 0) return AwFmSuccess;
  }
  if (g->amino && !dOffsets) { /* large fixed-length amino batches: the deeper table looked up first (awfm_amino_lookup_kernel.h) */
    DeviceGuard guard(g->device);
    const int did = awfmGpuAminoLookupSearch(g, (hipStream_t)stream, dChars, fixedLength, numQueries, (ulonglong2 *)dRanges, dCounts,
                                             rangesOfHitsOnly);
    if (did < 0 && did != -(int)AwFmAllocationFailure) return (enum AwFmReturnCode)(-did);
    if (did > 0) return AwFmSuccess;
  }
  return searchGeneral(g, dChars, dOffsets, fixedLength, numQueries, dRanges, dCounts, stream);
}

void awfmGpuIndexSetOrdered(AwFmGpuIndex *g, int mode) {
  if (g) g->orderMode = mode < 0 ? -1 : (mode != 0);
}

/* Instrumented run of the search kernel: tallyOut = {seeded queries, backward steps, distinct
 * blocks over those steps, query characters}.  Synchronous; not for timing. */
enum AwFmReturnCode awfmGpuSearchTally(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                       uint32_t fixedLength, uint64_t numQueries, uint64_t tallyOut[4]) {
  if (!g || !dChars || !tallyOut || (!dOffsets && fixedLength == 0)) {
    setError("awfmGpuSearchTally: null argument");
    return AwFmNullPtrError;
  }
  DeviceGuard guard(g->device);
  unsigned long long *dTally = nullptr;
  AWFM_HIP_TRY(hipMalloc((void **)&dTally, 32), AwFmAllocationFailure);
  hipError_t e = hipMemset(dTally, 0, 32);
  if (e == hipSuccess && numQueries) {
    /* the tally prices the reference algorithm (index seed table), so this launch gets a copy of
     * the image view without the device-only deeper table; the image itself is not touched (other threads may
     * be searching through it) */
    DevIndex plain = g->dev;
    if (!awfmGpuDiag("tally_with_deep")) { /* (diagnostics: what the kernel executes behind the deeper table) */
      plain.deepSeed = nullptr;
      plain.deepK = 0;
    }
    launchSearch<true>(g, plain, lanesPerQuery(g), (hipStream_t)0, dChars, (const unsigned long long *)dOffsets,
                       fixedLength, numQueries, nullptr, nullptr, dTally);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(tallyOut, dTally, 32, hipMemcpyDeviceToHost);
  (void)hipFree(dTally);
  if (e != hipSuccess) {
    setError("awfmGpuSearchTally", e);
    return AwFmGeneralFailure;
  }
  return AwFmSuccess;
}

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
  return launchLocate(g, totalHits, (unsigned long long *)dPositions, s, (unsigned long long *)outPositions);
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
  return launchLocate(g, capacityHits, (unsigned long long *)dPositions, s, (unsigned long long *)dPositions, total);
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
  return launchLocate(g, capacityHits, (unsigned long long *)dPositions, s, (unsigned long long *)dPositions,
                      (const unsigned long long *)dHitOffsets + capacity);
}

namespace {
/* LF-walk + sampled-SA kernel over `totalHits` BWT positions stored in dPositions (in place) */
enum AwFmReturnCode launchLocate(AwFmGpuIndex *g, unsigned long long totalHits, unsigned long long *dPositions,
                                 hipStream_t s, unsigned long long *out, const unsigned long long *totalOnDevice, unsigned stepCap) {
  {
    /* the walk runs at the rate the chip delivers random granules whatever the group width (17.2 / 17.5 / 18.3 ms
     * for g4 / g2 / g1 on 1.0007*10^8 hits); four lanes keep the fewest instructions per step */
    int lanes = g->kernel == AWFM_GPU_KERNEL_AUTO ? 4 : lanesPerQuery(g);
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
}  // namespace

/* Optional: the full suffix array on the device (32-bit entries), computed once with the LF-walk kernel
 * from the sampled SA, so that a locate becomes one gather.  enable = 0 drops it. */
enum AwFmReturnCode awfmGpuIndexSetDenseSa(AwFmGpuIndex *g, int enable) {
  if (!g) {
    setError("awfmGpuIndexSetDenseSa: null image");
    return AwFmNullPtrError;
  }
  if (g->shares) {
    setError("awfmGpuIndexSetDenseSa: set it on the primary image, not on a lane");
    return AwFmIllegalPositionError;
  }
  DeviceGuard guard(g->device);
  LaneLocks lanes(g);
  std::lock_guard<std::mutex> lock(g->workMutex);
  const enum AwFmReturnCode rc = applyDenseSa(g, enable != 0);
  for (AwFmGpuIndex *lane : lanes.lanes) {
    lane->dDenseSa = g->dDenseSa;
    lane->denseWide = g->denseWide;
  }
  return rc;
}

/* the caller holds whatever locks the image needs (none for an image nobody else has a pointer to yet) */
namespace {
constexpr unsigned kDenseUnknown = 0xFFFFFFFFu; /* an entry the capped walk did not reach a sample for (no position: n < 2^32 - 1) */
/* a chunk of the construction: final positions to 32 bits; a parked walk (kWalkParked) leaves kDenseUnknown and, in `park`,
 * {steps walked << 32 | the position it stands at}: SA[this] = SA[that position] + steps */
__global__ void __launch_bounds__(256) narrowParkKernel(const unsigned long long *__restrict__ in, unsigned long long count,
                                                        unsigned *__restrict__ dense, unsigned long long *__restrict__ park,
                                                        unsigned long long *__restrict__ parked) {
  unsigned long long mine = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * 256ull) {
    const unsigned long long v = in[i];
    if (v & kWalkParked) {
      dense[i] = kDenseUnknown;
      if (park) park[i] = (((v >> 40) & 0x3FFFFFull) << 32) | (v & 0xFFFFFFFFull); /* (NULL: the pass that only counts) */
      mine++;
    } else {
      dense[i] = (unsigned)v;
    }
  }
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
  if ((threadIdx.x & 63u) == 0u && mine) atomicAdd(parked, mine);
}
/* One round of completing the parked entries from each other: entry j = {d, t} says SA[j] = SA[t] + d (mod n).  When t is
 * known by now, so is j; otherwise j takes t's own {d', t'} on board -- SA[j] = SA[t'] + d + d' -- which at least doubles the
 * distance it looks ahead every round (pointer jumping along the LF permutation; an entry read while another thread rewrites
 * it is valid before and after: 8-byte loads and stores).  `left`: entries still unknown after the round. */
__global__ void __launch_bounds__(256) denseSaJumpKernel(unsigned *__restrict__ dense, unsigned long long *__restrict__ park,
                                                         unsigned long long n, unsigned long long *__restrict__ left) {
  unsigned long long mine = 0;
  for (unsigned long long j = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; j < n; j += (unsigned long long)gridDim.x * 256ull) {
    if (dense[j] != kDenseUnknown) continue;
    const unsigned long long e = ((volatile unsigned long long *)park)[j];
    const unsigned t = (unsigned)e;
    const unsigned long long d = e >> 32;
    const unsigned at = ((volatile unsigned *)dense)[t];
    if (at != kDenseUnknown) {
      dense[j] = (unsigned)(((unsigned long long)at + d) % n);
    } else {
      const unsigned long long e2 = ((volatile unsigned long long *)park)[t];
      park[j] = ((d + (e2 >> 32)) << 32) | (e2 & 0xFFFFFFFFull);
      mine++;
    }
  }
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
  if ((threadIdx.x & 63u) == 0u && mine) atomicAdd(left, mine);
}
/* Round 5: the same with the parked walks in a LIST.  The first pass over a chunk appends a parked walk's {position j, {steps,
 * where it stands}} to the list and leaves the entry's SLOT in dense[j]; nothing of 8 bytes per position is allocated and no
 * position is walked twice (a genome-shaped 3.1 Gbp text parks 3.6 * 10^7 of its walks: 0.4 GB of list instead of 24.8 GB).
 * Slots beyond the list's capacity are only counted: the caller then takes the array of all positions above. */
__global__ void __launch_bounds__(256) narrowParkListKernel(const unsigned long long *__restrict__ in, unsigned long long count,
                                                            unsigned long long first, unsigned *__restrict__ dense,
                                                            unsigned *__restrict__ listAt, unsigned long long *__restrict__ listEntry,
                                                            unsigned long long capacity, unsigned long long *__restrict__ parked) {
  const unsigned lane = threadIdx.x & 63u;
  for (unsigned long long at = (unsigned long long)blockIdx.x * 256ull; at < count; at += (unsigned long long)gridDim.x * 256ull) {
    const unsigned long long i = at + threadIdx.x;
    const unsigned long long v = i < count ? in[i] : 0ull;
    const bool isParked = (v & kWalkParked) != 0ull;
    const unsigned long long mask = __ballot(isParked);
    if (mask == 0ull) {
      if (i < count) dense[first + i] = (unsigned)v;
      continue;
    }
    unsigned long long base = 0;
    const unsigned leader = (unsigned)__ffsll((long long)mask) - 1u;
    if (lane == leader) base = atomicAdd(parked, (unsigned long long)__popcll(mask));
    base = __shfl(base, (int)leader);
    if (isParked) {
      const unsigned long long slot = base + (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
      if (slot < capacity) {
        listAt[slot] = (unsigned)(first + i);
        listEntry[slot] = (((v >> 40) & 0x3FFFFFull) << 32) | (v & 0xFFFFFFFFull);
        dense[first + i] = (unsigned)slot;
      } else {
        dense[first + i] = kDenseUnknown;
      }
    } else if (i < count) {
      dense[first + i] = (unsigned)v;
    }
  }
}
/* One round over the list.  Slot s is still open while dense[its position] == s.  Whether the position t it waits for is known
 * is read off dense[t] alone: a value x with x < listed and listAt[x] == t is t's slot -- or, once in 2^32 or so, t's final
 * position that happens to equal its slot number; t is then taken for open, which is harmless: its entry {d', t'} stays a true
 * statement about SA[t] for ever, so j takes it on board and gets its answer from further along the walk (a chain ends at a
 * position that was never parked, and those are always recognised).  The same goes for a stale dense[t] from another XCD's
 * L2.  A finished slot whose value equals its number is computed again every round, to the same value. */
__global__ void __launch_bounds__(256) denseSaJumpListKernel(unsigned *__restrict__ dense, const unsigned *__restrict__ listAt,
                                                             unsigned long long *__restrict__ listEntry, unsigned long long listed,
                                                             unsigned long long n, unsigned long long *__restrict__ left) {
  unsigned long long mine = 0;
  for (unsigned long long s = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; s < listed; s += (unsigned long long)gridDim.x * 256ull) {
    const unsigned j = listAt[s];
    if ((unsigned long long)dense[j] != s) continue;
    const unsigned long long e = listEntry[s];
    const unsigned t = (unsigned)e;
    const unsigned long long d = e >> 32;
    const unsigned at = ((volatile unsigned *)dense)[t];
    const bool open = (unsigned long long)at < listed && listAt[at] == t;
    if (!open) {
      dense[j] = (unsigned)(((unsigned long long)at + d) % n);
    } else {
      const unsigned long long e2 = ((volatile unsigned long long *)listEntry)[at];
      listEntry[s] = ((d + (e2 >> 32)) << 32) | (e2 & 0xFFFFFFFFull);
      mine++;
    }
  }
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
  if ((threadIdx.x & 63u) == 0u && mine) atomicAdd(left, mine);
}
/* ---- the same construction for images of 2^32 positions and more (round 6; ref src/AwFmSuffixArray.c:12-18 is 64-bit) ----
 * The array is put together in 64-bit entries and packed to 40 bits at the end (DenseSa).  An entry that is still open holds
 * bit 63 and the slot of its parked walk in the list -- no value can be mistaken for one --, a list entry is {steps so far,
 * the position the walk stands at}, and a round reads the entries the round before wrote (two copies of the list), so that a
 * 16-byte entry is never read while it is rewritten. */
constexpr unsigned long long kDenseOpen = 1ull << 63;
__global__ void __launch_bounds__(256) parkWideKernel(const unsigned long long *__restrict__ in, unsigned long long count, unsigned long long first,
                                                      unsigned long long *__restrict__ dense, unsigned long long *__restrict__ listAt,
                                                      ulonglong2 *__restrict__ listEntry, unsigned long long capacity,
                                                      unsigned long long *__restrict__ parked) {
  const unsigned lane = threadIdx.x & 63u;
  for (unsigned long long at = (unsigned long long)blockIdx.x * 256ull; at < count; at += (unsigned long long)gridDim.x * 256ull) {
    const unsigned long long i = at + threadIdx.x;
    const unsigned long long v = i < count ? in[i] : 0ull;
    const bool isParked = (v & kWalkParked) != 0ull;
    const unsigned long long mask = __ballot(isParked);
    if (mask == 0ull) {
      if (i < count) dense[first + i] = v;
      continue;
    }
    unsigned long long base = 0;
    const unsigned leader = (unsigned)__ffsll((long long)mask) - 1u;
    if (lane == leader) base = atomicAdd(parked, (unsigned long long)__popcll(mask));
    base = __shfl(base, (int)leader);
    if (isParked) {
      const unsigned long long slot = base + (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
      if (slot < capacity) {
        listAt[slot] = first + i;
        listEntry[slot] = make_ulonglong2((v >> 40) & 0x3FFFFFull, v & kWalkSampleMask);
      }
      dense[first + i] = kDenseOpen | slot; /* (beyond the capacity: the caller sees the count and starts over) */
    } else if (i < count) {
      dense[first + i] = v;
    }
  }
}
__global__ void __launch_bounds__(256) denseSaJumpWideKernel(unsigned long long *__restrict__ dense, const unsigned long long *__restrict__ listAt,
                                                             const ulonglong2 *__restrict__ entryIn, ulonglong2 *__restrict__ entryOut,
                                                             unsigned long long listed, unsigned long long n, unsigned long long *__restrict__ left) {
  unsigned long long mine = 0;
  for (unsigned long long s = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; s < listed; s += (unsigned long long)gridDim.x * 256ull) {
    const unsigned long long j = listAt[s];
    const ulonglong2 e = entryIn[s];
    entryOut[s] = e;
    if ((dense[j] & kDenseOpen) == 0ull) continue;
    const unsigned long long at = ((volatile unsigned long long *)dense)[e.y];
    if ((at & kDenseOpen) == 0ull) {
      dense[j] = (at + e.x) % n;
    } else {
      const ulonglong2 e2 = entryIn[at & ~kDenseOpen];
      entryOut[s] = make_ulonglong2(e.x + e2.x, e2.y);
      mine++;
    }
  }
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
  if ((threadIdx.x & 63u) == 0u && mine) atomicAdd(left, mine);
}
}  // namespace

/* the full suffix array an index builder of this thread still holds (awfm_gpu_build.hip: 32-bit positions of the text it has
 * just sorted): the image it adopts next takes it as it is instead of walking every position to its sample */
extern "C++" {
thread_local void *awfmGpuDenseSaStash = nullptr;
thread_local unsigned long long awfmGpuDenseSaStashLength = 0;
thread_local bool awfmGpuDenseSaStashWide = false;
}

/* capped (the AUTOMATIC construction): a position that has not reached a sample after 32 x ratio LF steps (a random walk is
 * that long once in e^32 positions) is parked where it stands, and the parked entries are completed from each other by
 * pointer jumping (denseSaJumpKernel: log2 of the longest chain rounds).  A text with R long runs of one letter, R a
 * multiple of the ratio (a genome's runs of N), otherwise costs the construction 10^5..10^7 steps for every position
 * inside a run: 566 s instead of 0.3 for the genome-shaped 3.1 Gbp text of bench.py --text repetitive.  The parked walks of
 * the one pass are kept in a list (12 bytes each); only a text that parks more than a quarter of its positions (or 2^26) pays
 * 8 bytes per position and a second pass.  Without memory for either, or when 64 rounds do not finish, no array is kept and
 * the image locates by walking, as the reference does; a construction that was asked for (awfmGpuIndexSetDenseSa,
 * $AWFM_GPU_DENSE_SA=1) then walks every position to its sample, however long that takes. */
static enum AwFmReturnCode applyDenseSaWide(AwFmGpuIndex *g, bool capped);
static enum AwFmReturnCode applyDenseSa(AwFmGpuIndex *g, bool enable, bool capped) {
  (void)hipDeviceSynchronize();
  if (g->dDenseSa) (void)hipFree(g->dDenseSa);
  g->dDenseSa = nullptr;
  g->denseSaBytes = 0;
  g->denseWide = false;
  if (!enable) return AwFmSuccess;
  const unsigned long long n = g->dev.bwtLength;
  /* 32-bit entries for the images that run 32-bit positions, 40-bit ones (DenseSa) for the others */
  const bool wide = !awfmImageNarrow(g);
  if (n >= (1ull << 40)) {
    setError("awfmGpuIndexSetDenseSa: 40-bit entries need bwtLength < 2^40");
    return AwFmUnsupportedVersionError;
  }
  if (awfmGpuDenseSaStash && awfmGpuDenseSaStashLength == n) { /* this thread's builder hands its array over */
    void *stash = awfmGpuDenseSaStash;
    const bool stashWide = awfmGpuDenseSaStashWide;
    awfmGpuDenseSaStash = nullptr;
    awfmGpuDenseSaStashLength = 0;
    if (stashWide != wide) { /* (tests: a small image forced wide, or a small text sorted with 64-bit positions) */
      void *other = nullptr;
      if (hipMalloc(&other, awfmDenseSaBytes(n, wide)) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(stash);
        return AwFmSuccess; /* no array: the image locates by walking */
      }
      if (wide) {
        hipLaunchKernelGGL((packDense40Kernel<unsigned>), dim3((unsigned)g->numCUs * 8u), dim3(256), 0, 0, (const unsigned *)stash, n, (unsigned *)other);
      } else {
        DenseSa from;
        from.words = (const unsigned *)stash;
        from.wide = 1u;
        hipLaunchKernelGGL(unpackDense40Kernel, dim3((unsigned)g->numCUs * 8u), dim3(256), 0, 0, from, n, (unsigned *)other);
      }
      const bool ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
      (void)hipFree(stash);
      if (!ok) {
        (void)hipFree(other);
        setError("awfmGpuIndexSetDenseSa: converting the builder's suffix array failed");
        return AwFmGeneralFailure;
      }
      stash = other;
    }
    g->dDenseSa = stash;
    g->denseWide = wide;
    g->denseSaBytes = awfmDenseSaBytes(n, wide);
    return AwFmSuccess;
  }
  if (wide) return applyDenseSaWide(g, capped);
  unsigned *dense = nullptr;
  unsigned long long *chunkBuf = nullptr, *park = nullptr, *counter = nullptr;
  const unsigned long long chunk = n < (1ull << 28) ? n : (1ull << 28);
  AWFM_HIP_TRY(hipMalloc((void **)&dense, n * 4), AwFmAllocationFailure);
  if (hipMalloc((void **)&chunkBuf, chunk * 8 + 16) != hipSuccess) {
    (void)hipFree(dense);
    setError("awfmGpuIndexSetDenseSa: hipMalloc of the work buffer failed");
    return AwFmAllocationFailure;
  }
  counter = chunkBuf + chunk; /* two words behind the chunk: parked entries, entries left */
  enum AwFmReturnCode rc = AwFmSuccess;
  /* every construction caps its walks and completes the parked ones by pointer jumping (round 5: the explicit one as well --
   * awfmGpuIndexSetDenseSa, $AWFM_GPU_DENSE_SA=1 -- which used to walk every position to the end: 566 s for a text with long
   * runs); `capped` = false now only says what happens when the parked walks cannot be kept: the array that was asked for
   * is then built by walking to the end, the automatic one is dropped */
  const bool explicitBuild = !capped;
  unsigned stepCap = 32u * g->dev.saRatio;
  /* the parked walks of the first pass go into a list (narrowParkListKernel) of at most a quarter of the positions, 2^26 at
   * most (0.8 GB; $AWFM_GPU_DIAG park_list = entries, 0 = none: tests): a text that parks more -- one that is mostly runs -- takes
   * the array over all positions and a second pass, as round 4 did for every text that parked anything */
  unsigned *listAt = nullptr;
  unsigned long long *listEntry = nullptr;
  unsigned long long listCapacity = n / 4u + 1024u < (1ull << 26) ? n / 4u + 1024u : (1ull << 26);
  if (const char *env = awfmGpuDiag("park_list")) listCapacity = strtoull(env, nullptr, 10);
  if (listCapacity > n) listCapacity = n;
  if (listCapacity != 0 && (hipMalloc((void **)&listAt, listCapacity * 4) != hipSuccess ||
                            hipMalloc((void **)&listEntry, listCapacity * 8) != hipSuccess)) {
    (void)hipGetLastError();
    if (listAt) (void)hipFree(listAt);
    listAt = nullptr;
    listEntry = nullptr;
    listCapacity = 0;
  }
  bool listing = listCapacity != 0;
  auto walkAll = [&]() { /* every position walked (capped: parked walks counted, and kept where there is a `park`) */
    if (hipMemset(counter, 0, 16) != hipSuccess) rc = AwFmGeneralFailure;
    for (unsigned long long first = 0; first < n && rc == AwFmSuccess; first += chunk) {
      const unsigned long long count = n - first < chunk ? n - first : chunk;
      hipLaunchKernelGGL(iotaKernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, chunkBuf, first, count);
      rc = launchLocate(g, count, chunkBuf, (hipStream_t)0, nullptr, nullptr, stepCap);
      if (stepCap && listing)
        hipLaunchKernelGGL(narrowParkListKernel, dim3((unsigned)g->numCUs * 8u), dim3(256), 0, 0, (const unsigned long long *)chunkBuf,
                           count, first, dense, listAt, listEntry, listCapacity, counter);
      else if (stepCap)
        hipLaunchKernelGGL(narrowParkKernel, dim3((unsigned)g->numCUs * 8u), dim3(256), 0, 0, (const unsigned long long *)chunkBuf, count,
                           dense + first, park ? park + first : (unsigned long long *)nullptr, counter);
      else
        hipLaunchKernelGGL(narrowKernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, chunkBuf, count, dense + first);
      if (hipGetLastError() != hipSuccess) rc = AwFmGeneralFailure;
    }
    if (hipDeviceSynchronize() != hipSuccess) rc = AwFmGeneralFailure;
  };
  walkAll();
  bool jumping = rc == AwFmSuccess;
  if (jumping) {
    unsigned long long parked = 0, left = 0;
    if (hipMemcpy(&parked, counter, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = AwFmGeneralFailure;
    const bool listed = listing && parked <= listCapacity; /* every parked walk of the one pass is in the list */
    listing = false;
    if (!listed && listAt) { /* (make room for the array over all positions) */
      (void)hipFree(listAt);
      (void)hipFree(listEntry);
      listAt = nullptr;
      listEntry = nullptr;
    }
    if (rc == AwFmSuccess && parked != 0 && !listed) {
      /* (the usual text parks nothing and never pays for this: 8 bytes per position, and the walks once more to fill them) */
      if (hipMalloc((void **)&park, n * 8) != hipSuccess) { /* no room to park walks */
        (void)hipGetLastError();
        park = nullptr;
        if (!explicitBuild) { /* no automatic array */
          (void)hipFree(chunkBuf);
          (void)hipFree(dense);
          return AwFmSuccess;
        }
        stepCap = 0u; /* the array was asked for: every walk to its sample, however long (exact: finishKernel resumes) */
        walkAll();
        jumping = false;
      } else {
        walkAll();
        if (rc == AwFmSuccess && hipMemcpy(&parked, counter, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = AwFmGeneralFailure;
      }
    }
    left = jumping ? parked : 0;
    unsigned rounds = 0;
    for (; rc == AwFmSuccess && left != 0 && rounds < 64u; rounds++) {
      if (hipMemset(counter + 1, 0, 8) != hipSuccess) rc = AwFmGeneralFailure;
      if (listed)
        hipLaunchKernelGGL(denseSaJumpListKernel, dim3((unsigned)g->numCUs * 8u), dim3(256), 0, 0, dense, (const unsigned *)listAt, listEntry,
                           parked, n, counter + 1);
      else
        hipLaunchKernelGGL(denseSaJumpKernel, dim3((unsigned)g->numCUs * 8u), dim3(256), 0, 0, dense, park, n, counter + 1);
      if (hipGetLastError() != hipSuccess || hipMemcpy(&left, counter + 1, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = AwFmGeneralFailure;
    }
    if (getenv("AWFM_VERBOSE") && parked)
      fprintf(stderr, "[awfm full suffix array] %llu of %llu walks parked after %u LF steps (%s); %u rounds of pointer jumping, %llu left\n",
              parked, n, stepCap, listed ? "in a list" : "an entry per position, walked twice", rounds, left);
    if (rc == AwFmSuccess && left != 0) { /* (64 rounds look 2^64 steps ahead: not reached by an index that is one) */
      if (listAt) (void)hipFree(listAt);
      if (listEntry) (void)hipFree(listEntry);
      if (park) (void)hipFree(park);
      (void)hipFree(chunkBuf);
      (void)hipFree(dense);
      return AwFmSuccess;
    }
  }
  if (park) (void)hipFree(park);
  if (listAt) (void)hipFree(listAt);
  if (listEntry) (void)hipFree(listEntry);
  (void)hipFree(chunkBuf);
  if (rc != AwFmSuccess) {
    (void)hipFree(dense);
    setError("awfmGpuIndexSetDenseSa: construction failed");
    return rc;
  }
  g->dDenseSa = dense;
  g->denseSaBytes = n * 4;
  return AwFmSuccess;
}

/* the construction for images that run 64-bit positions (kernels above): capped walks, the parked ones in a list, pointer
 * jumping, 40-bit entries at the end.  A text that parks more walks than the list holds -- a quarter of its positions, 2^27 at
 * most -- gets no automatic array; one that was asked for is then walked to the end, however long that takes. */
static enum AwFmReturnCode applyDenseSaWide(AwFmGpuIndex *g, bool capped) {
  const unsigned long long n = g->dev.bwtLength;
  const unsigned long long chunk = n < (1ull << 28) ? n : (1ull << 28);
  unsigned long long *dense = nullptr, *chunkBuf = nullptr, *listAt = nullptr;
  ulonglong2 *entry[2] = {nullptr, nullptr};
  unsigned long long listCapacity = n / 4u + 1024u < (1ull << 27) ? n / 4u + 1024u : (1ull << 27);
  auto release = [&]() {
    if (dense) (void)hipFree(dense);
    if (chunkBuf) (void)hipFree(chunkBuf);
    if (listAt) (void)hipFree(listAt);
    if (entry[0]) (void)hipFree(entry[0]);
    if (entry[1]) (void)hipFree(entry[1]);
    dense = chunkBuf = listAt = nullptr;
    entry[0] = entry[1] = nullptr;
  };
  if (hipMalloc((void **)&dense, n * 8) != hipSuccess || hipMalloc((void **)&chunkBuf, chunk * 8 + 16) != hipSuccess ||
      hipMalloc((void **)&listAt, listCapacity * 8) != hipSuccess || hipMalloc((void **)&entry[0], listCapacity * 16) != hipSuccess ||
      hipMalloc((void **)&entry[1], listCapacity * 16) != hipSuccess) {
    (void)hipGetLastError();
    release();
    setError("awfmGpuIndexSetDenseSa: no device memory for the construction");
    return capped ? AwFmSuccess : AwFmAllocationFailure;
  }
  unsigned long long *counter = chunkBuf + chunk; /* two words behind the chunk: parked entries, entries left */
  enum AwFmReturnCode rc = AwFmSuccess;
  unsigned stepCap = 32u * g->dev.saRatio;
  auto walkAll = [&]() {
    if (hipMemset(counter, 0, 16) != hipSuccess) rc = AwFmGeneralFailure;
    for (unsigned long long first = 0; first < n && rc == AwFmSuccess; first += chunk) {
      const unsigned long long count = n - first < chunk ? n - first : chunk;
      hipLaunchKernelGGL(iotaKernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, chunkBuf, first, count);
      rc = launchLocate(g, count, chunkBuf, (hipStream_t)0, nullptr, nullptr, stepCap);
      if (stepCap)
        hipLaunchKernelGGL(parkWideKernel, dim3((unsigned)g->numCUs * 8u), dim3(256), 0, 0, (const unsigned long long *)chunkBuf, count, first, dense,
                           listAt, entry[0], listCapacity, counter);
      else if (hipMemcpyAsync(dense + first, chunkBuf, count * 8, hipMemcpyDeviceToDevice, (hipStream_t)0) != hipSuccess)
        rc = AwFmGeneralFailure;
      if (hipGetLastError() != hipSuccess) rc = AwFmGeneralFailure;
    }
    if (hipDeviceSynchronize() != hipSuccess) rc = AwFmGeneralFailure;
  };
  walkAll();
  unsigned long long parked = 0, left = 0;
  if (rc == AwFmSuccess && hipMemcpy(&parked, counter, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = AwFmGeneralFailure;
  if (rc == AwFmSuccess && parked > listCapacity) {
    if (capped) { /* no automatic array for such a text */
      release();
      return AwFmSuccess;
    }
    stepCap = 0u; /* asked for: every walk to its sample (exact: finishKernel resumes the ones the walk kernel gives up) */
    walkAll();
    parked = 0;
  }
  left = parked;
  unsigned rounds = 0;
  for (; rc == AwFmSuccess && left != 0 && rounds < 64u; rounds++) {
    if (hipMemset(counter + 1, 0, 8) != hipSuccess) rc = AwFmGeneralFailure;
    hipLaunchKernelGGL(denseSaJumpWideKernel, dim3((unsigned)g->numCUs * 8u), dim3(256), 0, 0, dense, (const unsigned long long *)listAt,
                       (const ulonglong2 *)entry[rounds & 1u], entry[(rounds & 1u) ^ 1u], parked, n, counter + 1);
    if (hipGetLastError() != hipSuccess || hipMemcpy(&left, counter + 1, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = AwFmGeneralFailure;
  }
  if (getenv("AWFM_VERBOSE") && parked)
    fprintf(stderr, "[awfm full suffix array, 40-bit entries] %llu of %llu walks parked after %u LF steps; %u rounds of pointer jumping, %llu left\n", parked, n,
            stepCap, rounds, left);
  if (rc == AwFmSuccess && left != 0) { /* (64 rounds look 2^64 steps ahead: not reached by an index that is one) */
    release();
    return AwFmSuccess;
  }
  (void)hipFree(chunkBuf);
  (void)hipFree(listAt);
  (void)hipFree(entry[0]);
  (void)hipFree(entry[1]);
  chunkBuf = listAt = nullptr;
  entry[0] = entry[1] = nullptr;
  unsigned *packed = nullptr;
  if (rc == AwFmSuccess && hipMalloc((void **)&packed, awfmDenseSaBytes(n, true)) != hipSuccess) {
    (void)hipGetLastError();
    release();
    setError("awfmGpuIndexSetDenseSa: no device memory for the array");
    return capped ? AwFmSuccess : AwFmAllocationFailure;
  }
  if (rc == AwFmSuccess) {
    hipLaunchKernelGGL((packDense40Kernel<unsigned long long>), dim3((unsigned)g->numCUs * 8u), dim3(256), 0, 0, (const unsigned long long *)dense, n, packed);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = AwFmGeneralFailure;
  }
  release();
  if (rc != AwFmSuccess) {
    if (packed) (void)hipFree(packed);
    setError("awfmGpuIndexSetDenseSa: construction failed");
    return rc;
  }
  g->dDenseSa = packed;
  g->denseWide = true;
  g->denseSaBytes = awfmDenseSaBytes(n, true);
  return AwFmSuccess;
}

/* $AWFM_GPU_DENSE_SA=0|1 on an image that was just created or adopted (no lanes, nobody else holds it); unset: automatic.
 * Automatic: an image beyond the caches (>= 2^26 positions, below 2^32: 32-bit entries) whose suffix array is sampled
 * gets the full one when four times its size is free on the device -- 12.4 GB of 288 for a GRCh38-sized image, computed
 * by the LF-walk kernel itself from the sampled array (0.3 s).  A locate is then one gather per hit instead of a chain
 * of ~ratio dependent block reads plus the sample: 10^8 planted 21-mers 18.1 -> 9.7 ms per step, and the longest chain of
 * a small batch (60 us) is gone.  Positions are those of the walk (it wrote them); the host index, its sampled array
 * and the .awfmi file are untouched. */
static enum AwFmReturnCode applyDenseSaFromEnv(AwFmGpuIndex *g) {
  bool want = false, automatic = false;
  const char *env = getenv("AWFM_GPU_DENSE_SA");
  if (env && !strcmp(env, "auto")) { /* the automatic construction whatever the image's size (tests) */
    want = automatic = g->dev.saRatio > 1u;
  } else if (env) {
    want = atoi(env) != 0;
  } else if (g->dev.bwtLength >= (1ull << 26) && g->dev.bwtLength < (1ull << 40) && g->dev.saRatio > 1u) {
    /* (round 5: from 2^26 positions instead of 2^28 -- a Swiss-Prot-sized amino image, 0.8 GB of entries: the LF walk of the
     * few hits of a shard's list was a chain of 130 us, a third of the shard's step) */
    size_t freeBytes = 0, totalBytes = 0;
    DeviceGuard guard(g->device);
    /* (32-bit entries; 40-bit ones, put together in 64-bit entries, for the images that run 64-bit positions: round 6) */
    const uint64_t entryBytes = awfmImageNarrow(g) ? 4u : (awfmGpuDenseSaStash && awfmGpuDenseSaStashLength == g->dev.bwtLength ? 5u : 8u);
    if (hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess) want = freeBytes / (awfmImageNarrow(g) ? 4u : 2u) >= g->dev.bwtLength * entryBytes + (1ull << 31);
    else (void)hipGetLastError();
    if (!want) g->accelNotes += "full suffix array: not built (less than 4 x its size free); ";
    automatic = true;
  }
  if (!want || g->dev.bwtLength >= (1ull << 40)) return AwFmSuccess;
  DeviceGuard guard(g->device);
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  const enum AwFmReturnCode rc = applyDenseSa(g, true, automatic);
  if (!g->dDenseSa) g->accelNotes += "full suffix array: not built (no device memory, or walks that could not be completed); ";
  clock_gettime(CLOCK_MONOTONIC, &t1);
  g->denseSaBuildSeconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  return rc;
}
int awfmGpuIndexHasDenseSa(const AwFmGpuIndex *g) { return g && g->dDenseSa ? 1 : 0; }
/* see include/awfm_gpu.h */
int awfmGpuIndexDescribe(const AwFmGpuIndex *g, char *out, int outBytes) {
  if (!g || !out || outBytes <= 0) return 0;
  const AwFmGpuIndex *p = g->shares ? g->shares : g;
  std::string text = std::string(p->amino ? "amino" : "nucleotide") + " image of " + std::to_string(p->dev.bwtLength) + " positions, " +
                     std::to_string(awfmGpuIndexDeviceBytes(p)) + " bytes on device " + std::to_string(p->device) + ": ";
  if (!p->amino) text += p->dev.pairBlocks ? "pair image yes; " : "pair image no; ";
  text += p->dev.deepK ? "deeper table depth " + std::to_string(p->dev.deepK) + (p->dev.deepNext ? " with next-step bits; " : "; ") : "deeper table no; ";
  text += p->dDenseSa ? "full suffix array yes; " : "full suffix array no; ";
  if (!p->amino) text += p->dLengthTable ? "tables per k-mer length 1.." + std::to_string(p->lengthDepths) + "; " : "tables per k-mer length not built (the first large mixed-length batch builds them); ";
  if (!p->accelNotes.empty()) text += "notes: " + p->accelNotes;
  while (!text.empty() && (text.back() == ' ' || text.back() == ';')) text.pop_back();
  const int n = (int)text.size() < outBytes - 1 ? (int)text.size() : outBytes - 1;
  memcpy(out, text.data(), (size_t)n);
  out[n] = 0;
  return (int)text.size();
}
double awfmGpuIndexDenseSaBuildSeconds(const AwFmGpuIndex *g) { return g ? (g->shares ? g->shares : g)->denseSaBuildSeconds : 0.0; }
/* the tables per k-mer length a mixed-length batch builds on first use (awfm_gpu_ordered.hip: ensureLengthTables) */
uint64_t awfmGpuIndexLengthTableBytes(const AwFmGpuIndex *g) { return g ? (g->shares ? g->shares : g)->lengthTableBytes : 0; }
double awfmGpuIndexLengthTableBuildSeconds(const AwFmGpuIndex *g) { return g ? (g->shares ? g->shares : g)->lengthTableBuildSeconds : 0.0; }

/* ---- host-buffer entry points ---- */

namespace {
struct HostBatchLayout {
  size_t chars, offsets, ranges, counts, hitOffsets, scratch, total;
};

HostBatchLayout layoutFor(uint64_t n, uint64_t totalChars, bool hasOffsets, bool locate) {
  HostBatchLayout l{};
  size_t at = 0;
  l.chars = at;
  at = alignUp(at + (totalChars ? totalChars : 1), 256);
  l.offsets = at;
  at = alignUp(at + (hasOffsets ? (n + 1) * 8 : 0), 256);
  l.ranges = at;
  at = alignUp(at + n * 16, 256);
  l.counts = at;
  at = alignUp(at + n * 4, 256);
  l.hitOffsets = at;
  at = alignUp(at + (locate ? (n + 1) * 8 : 0), 256);
  l.scratch = at;
  at = alignUp(at + (locate ? awfmGpuScanScratchBytes(n) : 0), 256);
  l.total = at;
  return l;
}
}  // namespace

enum AwFmReturnCode awfmGpuCountHost(AwFmGpuIndex *g, const uint8_t *chars, const uint64_t *offsets,
                                     uint32_t fixedLength, uint64_t numQueries, struct AwFmSearchRange *ranges,
                                     uint32_t *counts) {
  if (!g || !chars) {
    setError("awfmGpuCountHost: null argument");
    return AwFmNullPtrError;
  }
  if (numQueries == 0) return AwFmSuccess;
  DeviceGuard guard(g->device);
  std::lock_guard<std::mutex> lock(g->workMutex);
  const uint64_t totalChars = offsets ? offsets[numQueries] : numQueries * (uint64_t)fixedLength;
  const HostBatchLayout l = layoutFor(numQueries, totalChars, offsets != nullptr, false);
  enum AwFmReturnCode rc = ensureWork(g, l.total);
  if (rc != AwFmSuccess) return rc;
  uint8_t *w = (uint8_t *)g->dWork;
  /* the calling thread's own stream: the host lanes (or two user threads with two images) overlap one's
   * transfers with the other's kernels */
  hipStream_t s = hipStreamPerThread;
  AWFM_HIP_TRY(hipMemcpyAsync(w + l.chars, chars, totalChars, hipMemcpyHostToDevice, s), AwFmGeneralFailure);
  if (offsets)
    AWFM_HIP_TRY(hipMemcpyAsync(w + l.offsets, offsets, (numQueries + 1) * 8, hipMemcpyHostToDevice, s), AwFmGeneralFailure);
  /* a caller that does not ask for the ranges only needs the hits (the AoS entry points): ordered search if it applies */
  rc = (ranges ? awfmGpuSearch : awfmGpuSearchHits)(g, w + l.chars, offsets ? (const uint64_t *)(w + l.offsets) : nullptr,
                                                    fixedLength, numQueries,
                                                    ranges ? (struct AwFmSearchRange *)(w + l.ranges) : nullptr,
                                                    (uint32_t *)(w + l.counts), s);
  if (rc != AwFmSuccess) return rc;
  if (ranges)
    AWFM_HIP_TRY(hipMemcpyAsync(ranges, w + l.ranges, numQueries * 16, hipMemcpyDeviceToHost, s), AwFmGeneralFailure);
  if (counts)
    AWFM_HIP_TRY(hipMemcpyAsync(counts, w + l.counts, numQueries * 4, hipMemcpyDeviceToHost, s), AwFmGeneralFailure);
  AWFM_HIP_TRY(hipStreamSynchronize(s), AwFmGeneralFailure);
  return AwFmSuccess;
}

/* How many hits' positions may be resident on the device at once: $AWFM_GPU_HIT_BUDGET_BYTES / 8, else a quarter of
 * the free device memory (what this image's own position buffer holds counted as free), between 2^25 and 2^31 hits. */
uint64_t awfmGpuHitBudget(const AwFmGpuIndex *g) {
  if (const char *env = getenv("AWFM_GPU_HIT_BUDGET_BYTES")) {
    const unsigned long long bytes = strtoull(env, nullptr, 10);
    if (bytes) return bytes / 8 > 1024 ? bytes / 8 : 1024;
  }
  /* the automatic budget: asked of the device once per image.  Up to three callers hold a budget's worth of positions at
   * a time (the AoS entry points' lanes, the pipeline's slots), each with page-locked staging of the same size: a quarter
   * of the free memory is shared among them, and 2^28 hits (2 GB) is as large as a window gets by itself */
  const uint64_t cached = __atomic_load_n(&g->hitBudgetAuto, __ATOMIC_RELAXED);
  if (cached) return cached;
  size_t freeBytes = 0, totalBytes = 0;
  if (hipMemGetInfo(&freeBytes, &totalBytes) != hipSuccess) {
    (void)hipGetLastError();
    freeBytes = (size_t)1 << 32;
  }
  uint64_t hits = ((uint64_t)freeBytes + g->hitsBytes) / 4 / 3 / 8;
  if (hits < (1ull << 25)) hits = 1ull << 25;
  if (hits > (1ull << 28)) hits = 1ull << 28;
  __atomic_store_n(&g->hitBudgetAuto, hits, __ATOMIC_RELAXED);
  return hits;
}

/* first query whose list ends after hit number h (offsets[q + 1] > h), i.e. the query hit h belongs to */
static uint64_t queryOfHit(const uint64_t *offsets, uint64_t n, uint64_t h) {
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    const uint64_t mid = lo + (hi - lo) / 2;
    if (offsets[mid + 1] > h) hi = mid;
    else lo = mid + 1;
  }
  return lo;
}

/* Upload, search, scan; then the hit list in windows of at most hitBudget() hits, two in flight: while the sink
 * consumes window w on the calling thread, the walk and the download of window w+1 run.  hitOffsets[0..n] is complete
 * before the first sink call. */
enum AwFmReturnCode awfmGpuLocateHostWindows(AwFmGpuIndex *g, const uint8_t *chars, const uint64_t *offsets,
                                             uint32_t fixedLength, uint64_t numQueries, struct AwFmSearchRange *ranges,
                                             uint64_t *hitOffsets, AwFmGpuHitWindowSink sink, void *user) {
  if (!g || !chars || !hitOffsets || !sink) {
    setError("awfmGpuLocateHost: null argument");
    return AwFmNullPtrError;
  }
  hitOffsets[0] = 0;
  if (numQueries == 0) return AwFmSuccess;
  DeviceGuard guard(g->device);
  std::lock_guard<std::mutex> lock(g->workMutex);
  const uint64_t totalChars = offsets ? offsets[numQueries] : numQueries * (uint64_t)fixedLength;
  const HostBatchLayout l = layoutFor(numQueries, totalChars, offsets != nullptr, true);
  enum AwFmReturnCode rc = ensureWork(g, l.total);
  if (rc != AwFmSuccess) return rc;
  uint8_t *w = (uint8_t *)g->dWork;
  hipStream_t s = hipStreamPerThread; /* see awfmGpuCountHost */
  AWFM_HIP_TRY(hipMemcpyAsync(w + l.chars, chars, totalChars, hipMemcpyHostToDevice, s), AwFmGeneralFailure);
  if (offsets)
    AWFM_HIP_TRY(hipMemcpyAsync(w + l.offsets, offsets, (numQueries + 1) * 8, hipMemcpyHostToDevice, s), AwFmGeneralFailure);
  struct AwFmSearchRange *dRanges = (struct AwFmSearchRange *)(w + l.ranges);
  uint64_t *dHitOffsets = (uint64_t *)(w + l.hitOffsets);
  rc = (ranges ? awfmGpuSearch : awfmGpuSearchHits)(g, w + l.chars, offsets ? (const uint64_t *)(w + l.offsets) : nullptr,
                                                    fixedLength, numQueries, dRanges, nullptr, s);
  if (rc != AwFmSuccess) return rc;
  uint64_t totalHits = 0;
  rc = awfmGpuHitOffsets(g, dRanges, numQueries, dHitOffsets, w + l.scratch, &totalHits, s);
  if (rc != AwFmSuccess) return rc;
  hipError_t e = hipMemcpyAsync(hitOffsets, dHitOffsets, (numQueries + 1) * 8, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess && ranges) e = hipMemcpyAsync(ranges, dRanges, numQueries * 16, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {
    setError("awfmGpuLocateHost: download of the hit offsets failed", e);
    return AwFmGeneralFailure;
  }
  if (totalHits == 0) return AwFmSuccess;
  /* one window when the list fits the budget (the usual case); otherwise windows of half the budget, two resident */
  const uint64_t budget = awfmGpuHitBudget(g);
  const uint64_t window = totalHits <= budget ? totalHits : (budget / 2 > 0 ? budget / 2 : 1);
  const unsigned buffers = totalHits <= budget ? 1u : 2u;
  if (window * buffers * 8 > g->hitsBytes) { /* grow-only, so that a steady stream of batches never allocates */
    if (g->dHits) (void)hipFree(g->dHits);
    g->dHits = nullptr;
    g->hitsBytes = 0;
    const size_t want = window * buffers * 8 + (buffers == 1 ? window * 2 : 0) + 4096;
    if (hipMalloc(&g->dHits, want) != hipSuccess) {
      (void)hipGetLastError();
      setError("awfmGpuLocateHost: no device memory for the positions of one window of hits ($AWFM_GPU_HIT_BUDGET_BYTES)");
      return AwFmAllocationFailure;
    }
    g->hitsBytes = want;
  }
  uint64_t *staging = (uint64_t *)awfmGpuPinnedBuffer(g, 3, window * buffers * 8);
  if (!staging) {
    setError("awfmGpuLocateHost: host allocation failed");
    return AwFmAllocationFailure;
  }
  if (!g->windowEvent[0]) {
    for (int i = 0; i < 2; i++)
      if (hipEventCreateWithFlags(&g->windowEvent[i], hipEventDisableTiming) != hipSuccess) {
        setError("awfmGpuLocateHost: hipEventCreate failed");
        return AwFmGeneralFailure;
      }
  }
  const uint64_t numWindows = (totalHits + window - 1) / window;
  struct Pending {
    uint64_t qb, qe, hb, he;
  } pending[2];
  auto issue = [&](uint64_t wi) -> enum AwFmReturnCode {
    Pending &p = pending[wi % buffers];
    p.hb = wi * window;
    p.he = p.hb + window < totalHits ? p.hb + window : totalHits;
    p.qb = queryOfHit(hitOffsets, numQueries, p.hb);
    p.qe = queryOfHit(hitOffsets, numQueries, p.he - 1) + 1;
    uint64_t *dPos = (uint64_t *)g->dHits + (wi % buffers) * window;
    const enum AwFmReturnCode r = awfmGpuLocateWindow(g, dRanges, dHitOffsets, p.qb, p.qe, p.hb, p.he, dPos, dPos, s);
    if (r != AwFmSuccess) return r;
    AWFM_HIP_TRY(hipMemcpyAsync(staging + (wi % buffers) * window, dPos, (p.he - p.hb) * 8, hipMemcpyDeviceToHost, s), AwFmGeneralFailure);
    AWFM_HIP_TRY(hipEventRecord(g->windowEvent[wi % buffers], s), AwFmGeneralFailure);
    return AwFmSuccess;
  };
  rc = issue(0);
  for (uint64_t wi = 0; wi < numWindows && rc == AwFmSuccess; wi++) {
    if (wi + 1 < numWindows) rc = issue(wi + 1); /* buffer (wi + 1) % 2 was handed to the sink an iteration ago */
    if (rc != AwFmSuccess) break;
    if (hipEventSynchronize(g->windowEvent[wi % buffers]) != hipSuccess) {
      setError("awfmGpuLocateHost: locate kernels failed", hipGetLastError());
      rc = AwFmGeneralFailure;
      break;
    }
    const Pending &p = pending[wi % buffers];
    if (sink(user, p.qb, p.qe, p.hb, p.he, staging + (wi % buffers) * window) != 0) {
      setError("awfmGpuLocateHost: the sink asked to stop");
      rc = AwFmGeneralFailure;
    }
  }
  (void)hipStreamSynchronize(s); /* nothing of this batch is in flight when the work buffers are released */
  return rc;
}

namespace {
struct FlatSinkCtx {
  uint64_t *positions;
};
int flatSink(void *user, uint64_t, uint64_t, uint64_t hitBegin, uint64_t hitEnd, const uint64_t *positions) {
  memcpy(((FlatSinkCtx *)user)->positions + hitBegin, positions, (hitEnd - hitBegin) * 8);
  return 0;
}
}  // namespace

enum AwFmReturnCode awfmGpuLocateHost(AwFmGpuIndex *g, const uint8_t *chars, const uint64_t *offsets,
                                      uint32_t fixedLength, uint64_t numQueries, struct AwFmSearchRange *ranges,
                                      uint64_t *hitOffsets, uint64_t **positions) {
  if (!positions) {
    setError("awfmGpuLocateHost: null argument");
    return AwFmNullPtrError;
  }
  *positions = nullptr;
  /* the flat result array needs the total first: a sink that allocates on its first window */
  struct Lazy {
    uint64_t *hitOffsets, n, *out;
    bool failed;
  } lazy = {hitOffsets, numQueries, nullptr, false};
  auto sink = [](void *user, uint64_t qb, uint64_t qe, uint64_t hb, uint64_t he, const uint64_t *pos) -> int {
    Lazy *z = (Lazy *)user;
    if (!z->out) {
      z->out = (uint64_t *)malloc((z->hitOffsets[z->n] ? z->hitOffsets[z->n] : 1) * 8);
      if (!z->out) {
        z->failed = true;
        return 1;
      }
    }
    FlatSinkCtx ctx = {z->out};
    return flatSink(&ctx, qb, qe, hb, he, pos);
  };
  const enum AwFmReturnCode rc = awfmGpuLocateHostWindows(g, chars, offsets, fixedLength, numQueries, ranges, hitOffsets, sink, &lazy);
  if (rc != AwFmSuccess || lazy.failed) {
    free(lazy.out);
    if (lazy.failed) setError("awfmGpuLocateHost: host allocation failed");
    return lazy.failed ? AwFmAllocationFailure : rc;
  }
  if (!lazy.out && hitOffsets) lazy.out = (uint64_t *)malloc(8); /* no hits: an empty array the caller can free */
  *positions = lazy.out;
  return AwFmSuccess;
}

}  // extern "C"
