/*
 * awfm_gpu_ordered.hip -- host side of the ordered hits-only search (awfm_ordered_kernel.h): decides whether a
 * batch takes it, owns its scratch memory in the device image, and launches
 *   fillNoHitKernel -> encodeQueriesKernel -> rocprim::radix_sort_pairs -> orderedSearchKernel -> searchKernel<INDIRECT>
 * on the caller's stream.  Called by awfmGpuSearchHits (awfm_gpu.hip); nothing here synchronises with the host
 * except a scratch (re)allocation.
 */
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <cstddef>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "awfm_ordered_kernel.h"

namespace {

inline size_t alignUp256(size_t v) { return (v + 255) / 256 * 256; }
constexpr size_t kOrderCounterBytes = 32768; /* 256 B for the count + 8 XCDs x 8 waves x 256 B of ticket counters */

struct OrderScratch {
  size_t keysIn, keysOut, recsIn, recsOut, generalCount, sortTemp, total;
};

OrderScratch scratchLayout(uint64_t n, size_t recordBytes, size_t sortTempBytes) {
  OrderScratch l;
  size_t at = 0;
  l.generalCount = at; /* word 0: the count; words 64, 128, ...: the ticket counters (8 XCDs x up to 8 waves) */
  at += kOrderCounterBytes;
  l.keysIn = at;
  at += alignUp256(n * sizeof(unsigned short));
  l.keysOut = at;
  at += alignUp256(n * sizeof(unsigned short));
  l.recsIn = at;
  at += alignUp256(n * recordBytes);
  l.recsOut = at;
  at += alignUp256(n * recordBytes);
  l.sortTemp = at;
  at += alignUp256(sortTempBytes ? sortTempBytes : 256);
  l.total = at;
  return l;
}

template <class Kernel>
unsigned residentGrid(const AwFmGpuIndex *g, Kernel kernel, size_t dynamicLds = 0, int threads = kThreads) {
  int perCU = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kernel, threads, dynamicLds) != hipSuccess || perCU < 1) perCU = 4;
  if (perCU > 8) perCU = 8;
  if (const char *env = getenv("AWFM_GPU_BLOCKS_PER_CU")) {
    const int v = atoi(env);
    if (v >= 1 && v <= 64) perCU = v;
  }
  return (unsigned)g->numCUs * (unsigned)perCU;
}

template <int G, bool NARROW, bool COMPACT, bool VARLEN, bool PAIR = false>
enum AwFmReturnCode launchOrderedKernel(AwFmGpuIndex *g, hipStream_t s, uint32_t len, unsigned depth, const ulonglong2 *table,
                                        unsigned long long nq, const void *recs, const unsigned short *keys,
                                        const unsigned *generalCount, ulonglong2 *rng, uint32_t *dCounts) {
  /* dynamic LDS: the 32-bit superblock bases of the pair image (images below 2^32 positions) */
  const bool superInLds = PAIR && NARROW && awfmPairSuperInLds(g);
  const size_t lds = superInLds ? (size_t)g->dev.numPairSuper * 64u : 0u; /* the 16 pair bases of every superblock */
  DevIndex dev = g->dev;
  dev.pairSuperInLds = superInLds ? 1u : 0u;
  constexpr int threads = orderedThreads(PAIR);
  unsigned grid = residentGrid(g, orderedSearchKernel<G, NARROW, COMPACT, VARLEN, PAIR>, lds, threads);
  const unsigned long long blocks = (nq + threads / G - 1) / (threads / G);
  if (blocks < grid) grid = (unsigned)blocks;
  if (grid >= 8u) grid &= ~7u; /* a multiple of the 8 XCDs, so that every XCD gets the same number of workgroups */
  /* measurement hook (bench.py): HIP events around the dominant kernel on its launch stream */
  const bool timed = getenv("AWFM_GPU_TIME_ORDERED") != nullptr;
  if (timed && !g->orderTiming[0]) {
    AWFM_HIP_TRY(hipEventCreate(&g->orderTiming[0]), AwFmGeneralFailure);
    AWFM_HIP_TRY(hipEventCreate(&g->orderTiming[1]), AwFmGeneralFailure);
  }
  if (timed) AWFM_HIP_TRY(hipEventRecord(g->orderTiming[0], s), AwFmGeneralFailure);
  hipLaunchKernelGGL((orderedSearchKernel<G, NARROW, COMPACT, VARLEN, PAIR>), dim3(grid ? grid : 1u), dim3(threads), lds, s, dev, recs,
                     keys, nq, generalCount, len, depth, table, rng, dCounts, (unsigned *)generalCount + 64,
                     getenv("AWFM_GPU_XCD_MAP") ? atoi(getenv("AWFM_GPU_XCD_MAP")) : 0);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  if (timed) AWFM_HIP_TRY(hipEventRecord(g->orderTiming[1], s), AwFmGeneralFailure);
  g->orderTimed = timed;
  return AwFmSuccess;
}

template <bool NARROW, bool COMPACT, bool VARLEN>
enum AwFmReturnCode launchOrdered(AwFmGpuIndex *g, hipStream_t s, const uint8_t *dChars, const unsigned long long *off,
                                  uint32_t len, unsigned depth, const ulonglong2 *table, unsigned long long nq,
                                  const void *recs, const unsigned short *keys, const unsigned *generalCount,
                                  ulonglong2 *rng, uint32_t *dCounts, bool packed = false) {
  {
    const char *lanes = getenv("AWFM_GPU_ORDERED_LANES"); /* measurement knob: 1 | 2 | 4 lanes per query (default 4) */
    const int G = lanes ? atoi(lanes) : 4;
    enum AwFmReturnCode rc;
    if (G == 4 && g->dev.pairBlocks && !getenv("AWFM_GPU_ORDERED_NO_PAIR"))
      rc = launchOrderedKernel<4, NARROW, COMPACT, VARLEN, true>(g, s, len, depth, table, nq, recs, keys, generalCount, rng, dCounts);
    else if (G == 2) rc = launchOrderedKernel<2, NARROW, COMPACT, VARLEN>(g, s, len, depth, table, nq, recs, keys, generalCount, rng, dCounts);
    else if (G == 1) rc = launchOrderedKernel<1, NARROW, COMPACT, VARLEN>(g, s, len, depth, table, nq, recs, keys, generalCount, rng, dCounts);
    else rc = launchOrderedKernel<4, NARROW, COMPACT, VARLEN>(g, s, len, depth, table, nq, recs, keys, generalCount, rng, dCounts);
    if (rc != AwFmSuccess) return rc;
  }
  if (packed) return AwFmSuccess; /* bit-packed k-mers: every one of them is covered */
  /* the queries the fast path left out (ambiguity characters, empty or longer than 32; normally none): general
   * kernel over the tail of the order */
  const unsigned grid = residentGrid(g, searchKernel<false, 4, VARLEN, false, NARROW, true>);
  hipLaunchKernelGGL((searchKernel<false, 4, VARLEN, false, NARROW, true>), dim3(grid), dim3(kThreads), 0, s, g->dev, dChars,
                     off, len, nq, rng, dCounts, (unsigned long long *)nullptr, (const unsigned char *)recs,
                     COMPACT ? 8u : (unsigned)sizeof(QueryRec), COMPACT ? 0u : (unsigned)offsetof(QueryRec, index), nq,
                     generalCount);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  return AwFmSuccess;
}

}  // namespace

/* does a batch of nq k-mers (fixed length, or CSR when hasOffsets) take the ordered path on this image?
 * depth/table: where the search of a fixed-length batch starts */
static bool orderedApplies(const AwFmGpuIndex *g, bool hasOffsets, uint32_t fixedLength, unsigned long long nq,
                           unsigned *depthOut, const ulonglong2 **tableOut) {
  if (g->amino || nq >= 0xFFFFFFFFull || g->dev.seedK == 0 || g->dev.seedK >= 32 || g->dev.deepK >= 32) return false;
  if (!hasOffsets) {
    if (fixedLength == 0 || fixedLength > 32) return false;
    /* the table the search starts from: the deeper device-only one when it is built and the k-mers reach it */
    const bool deep = g->dev.deepK != 0 && fixedLength >= g->dev.deepK;
    const unsigned depth = deep ? g->dev.deepK : g->dev.seedK;
    if (fixedLength < depth) return false; /* shorter than the seed: the general kernel */
    if (depthOut) *depthOut = depth;
    if (tableOut) *tableOut = deep ? g->dev.deepSeed : g->dev.seed;
  }
  int mode = g->orderMode; /* -1 auto, 0 off, 1 on */
  if (mode < 0) {
    if (const char *env = getenv("AWFM_GPU_ORDERED")) mode = atoi(env) != 0;
  }
  if (mode < 0) {
    /* worth its sort and its extra launches only when the batch is large and the image far exceeds the L2s.  Against
     * the general kernel with pair steps, 3.1 Gbp index, random / planted 21-mers (scripts/threshold_probe.sh): 4 M
     * k-mers 0.52 against 0.42 ms / 0.81 against 0.63 ms, 8 M 0.85 against 0.88 / 1.39 against 1.31, 16 M 1.45 against
     * 1.68 / 2.35 against 2.51, 64 M 4.44 against 6.34 / 7.93 against 9.85 */
    mode = nq >= (1ull << 23) && g->dev.bwtLength >= (1ull << 28);
  }
  return mode != 0;
}

/* milliseconds orderedSearchKernel took in the last awfmGpuSearchHits on this image that ran with
 * $AWFM_GPU_TIME_ORDERED set (waits for it); negative when there is none */
extern "C" double awfmGpuLastOrderedKernelMs(AwFmGpuIndex *g) {
  if (!g) return -1.0;
  std::lock_guard<std::mutex> lock(g->orderMutex);
  if (!g->orderTimed || !g->orderTiming[0]) return -1.0;
  float ms = 0.0f;
  if (hipEventSynchronize(g->orderTiming[1]) != hipSuccess ||
      hipEventElapsedTime(&ms, g->orderTiming[0], g->orderTiming[1]) != hipSuccess)
    return -1.0;
  return (double)ms;
}

extern "C" int awfmGpuSearchHitsIsOrdered(const AwFmGpuIndex *g, int hasOffsets, uint32_t fixedLength, uint64_t numQueries) {
  if (!g) return 0;
  if (!(g->kernel == AWFM_GPU_KERNEL_AUTO || g->kernel == AWFM_GPU_KERNEL_GROUP4)) return 0;
  return orderedApplies(g, hasOffsets != 0, fixedLength, numQueries, nullptr, nullptr) ? 1 : 0;
}

/* 1: the batch was searched; 0: the ordered path does not apply (caller runs the general kernel); <0: -AwFmReturnCode */
int awfmGpuOrderedSearch(AwFmGpuIndex *g, hipStream_t s, const uint8_t *dChars, const unsigned long long *off,
                         uint32_t fixedLength, unsigned long long nq, ulonglong2 *rng, uint32_t *dCounts, bool packed,
                         bool rangesOfHitsOnly) {
  unsigned depth = 0;
  const ulonglong2 *table = nullptr;
  if (packed && off) return 0;
  if (!orderedApplies(g, off != nullptr, fixedLength, nq, &depth, &table)) return 0;

  std::lock_guard<std::mutex> lock(g->orderMutex);
  /* 8-byte records: fixed-length batches of short enough k-mers */
  const bool compact = !off && orderCompact(fixedLength, depth) && !getenv("AWFM_GPU_ORDERED_WIDE");
  unsigned short *nullKeys = nullptr;
  size_t sortTemp = 0;
  hipError_t sized;
  if (compact) {
    unsigned long long *nullRecs = nullptr;
    sized = rocprim::radix_sort_pairs(nullptr, sortTemp, nullKeys, nullKeys, nullRecs, nullRecs, (size_t)nq, 0u, kOrderKeyBits, s);
  } else {
    QueryRec *nullRecs = nullptr;
    sized = rocprim::radix_sort_pairs(nullptr, sortTemp, nullKeys, nullKeys, nullRecs, nullRecs, (size_t)nq, 0u, kOrderKeyBits, s);
  }
  if (sized != hipSuccess) {
    setError("awfmGpuSearchHits: radix sort sizing failed");
    return -(int)AwFmGeneralFailure;
  }
  const OrderScratch l = scratchLayout(nq, compact ? 8 : sizeof(QueryRec), sortTemp);
  if (l.total > g->orderBytes) {
    /* hipFree waits for every stream of the device, so nothing still reads the old scratch */
    if (g->dOrder) (void)hipFree(g->dOrder);
    g->dOrder = nullptr;
    g->orderBytes = 0;
    const size_t want = l.total + l.total / 8;
    if (hipMalloc(&g->dOrder, want) != hipSuccess) {
      (void)hipGetLastError();
      return 0; /* no room for the scratch: the general kernel needs none */
    }
    g->orderBytes = want;
  }
  if (!g->orderEvent && hipEventCreateWithFlags(&g->orderEvent, hipEventDisableTiming) != hipSuccess) {
    setError("awfmGpuSearchHits: hipEventCreate failed");
    return -(int)AwFmGeneralFailure;
  }
  /* the scratch is shared by all searches on this image: order them across streams */
  if (g->orderEventRecorded && hipStreamWaitEvent(s, g->orderEvent, 0) != hipSuccess) {
    setError("awfmGpuSearchHits: hipStreamWaitEvent failed");
    return -(int)AwFmGeneralFailure;
  }
  uint8_t *w = (uint8_t *)g->dOrder;
  unsigned *generalCount = (unsigned *)(w + l.generalCount);
  unsigned short *keysIn = (unsigned short *)(w + l.keysIn), *keysOut = (unsigned short *)(w + l.keysOut);
  void *recsIn = w + l.recsIn, *recsOut = w + l.recsOut;
#define ORDER_TRY(call)                     \
  do {                                      \
    hipError_t e__ = (call);                \
    if (e__ != hipSuccess) {                \
      setError(#call, e__);                 \
      return -(int)AwFmGeneralFailure;      \
    }                                       \
  } while (0)
  ORDER_TRY(hipMemsetAsync(generalCount, 0, kOrderCounterBytes, s)); /* the count and the ticket counters */
  /* rangesOfHitsOnly (awfmGpuSearchHitsSparse): the counts say which k-mers have hits, so only the counts are
   * pre-filled and the ranges of the others stay as the caller left them -- 16 of the 20 bytes per k-mer not written */
  hipLaunchKernelGGL(fillNoHitKernel, dim3((unsigned)g->numCUs * 8u), dim3(256), 0, s,
                     rangesOfHitsOnly && dCounts ? (ulonglong2 *)nullptr : rng, dCounts, nq);
  ORDER_TRY(hipGetLastError());
  const unsigned encodeGrid = (unsigned)((nq + 255) / 256);
  const unsigned seedK = g->dev.seedK, deepK = g->dev.deepK;
  size_t tempBytes = sortTemp;
  if (compact) {
    if (packed)
      hipLaunchKernelGGL((encodeQueriesKernel<true, false, true>), dim3(encodeGrid), dim3(256), 0, s, dChars, off, fixedLength,
                         depth, seedK, deepK, nq, keysIn, recsIn, generalCount);
    else
      hipLaunchKernelGGL((encodeQueriesKernel<true, false>), dim3(encodeGrid), dim3(256), 0, s, dChars, off, fixedLength, depth,
                         seedK, deepK, nq, keysIn, recsIn, generalCount);
    ORDER_TRY(hipGetLastError());
    ORDER_TRY(rocprim::radix_sort_pairs(w + l.sortTemp, tempBytes, keysIn, keysOut, (unsigned long long *)recsIn,
                                        (unsigned long long *)recsOut, (size_t)nq, 0u, kOrderKeyBits, s));
  } else {
    if (off)
      hipLaunchKernelGGL((encodeQueriesKernel<false, true>), dim3(encodeGrid), dim3(256), 0, s, dChars, off, fixedLength, depth,
                         seedK, deepK, nq, keysIn, recsIn, generalCount);
    else if (packed)
      hipLaunchKernelGGL((encodeQueriesKernel<false, false, true>), dim3(encodeGrid), dim3(256), 0, s, dChars, off, fixedLength,
                         depth, seedK, deepK, nq, keysIn, recsIn, generalCount);
    else
      hipLaunchKernelGGL((encodeQueriesKernel<false, false>), dim3(encodeGrid), dim3(256), 0, s, dChars, off, fixedLength, depth,
                         seedK, deepK, nq, keysIn, recsIn, generalCount);
    ORDER_TRY(hipGetLastError());
    ORDER_TRY(rocprim::radix_sort_pairs(w + l.sortTemp, tempBytes, keysIn, keysOut, (QueryRec *)recsIn, (QueryRec *)recsOut,
                                        (size_t)nq, 0u, kOrderKeyBits, s));
  }
  const bool narrow = awfmImageNarrow(g);
  enum AwFmReturnCode rc;
#define ORDER_GO(NR, CP, VL) \
  launchOrdered<NR, CP, VL>(g, s, dChars, off, fixedLength, depth, table, nq, recsOut, keysOut, generalCount, rng, dCounts, packed)
  if (off) rc = narrow ? ORDER_GO(true, false, true) : ORDER_GO(false, false, true);
  else if (compact) rc = narrow ? ORDER_GO(true, true, false) : ORDER_GO(false, true, false);
  else rc = narrow ? ORDER_GO(true, false, false) : ORDER_GO(false, false, false);
#undef ORDER_GO
  if (rc != AwFmSuccess) return -(int)rc;
  ORDER_TRY(hipEventRecord(g->orderEvent, s));
  g->orderEventRecorded = true;
#undef ORDER_TRY
  return 1;
}
