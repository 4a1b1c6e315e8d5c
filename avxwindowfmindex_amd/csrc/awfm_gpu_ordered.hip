/*
 * awfm_gpu_ordered.hip -- host side of the hits-only searches of large batches (awfm_ordered_kernel.h and the lookup kernels):
 * decides which path a batch takes, owns the scratch memory in the device image (two slots, gated across streams), and
 * launches, on the caller's stream,
 *   lookupPrepKernel -> lookupSearchKernel -> searchKernel<INDIRECT>                                   (lookup first, predicted)
 *   ... -> encodeCodes4Kernel -> bucketScanSharesKernel -> partitionKernel -> orderedSearchKernel -> searchKernel<INDIRECT>
 * for fixed-length batches (8-byte records), the same passes over 16-byte records or mixedLookupSearchKernel for the others,
 * aminoLookupSearchKernel for amino batches, exactLookupSearchKernel behind awfmGpuSearch.  Called by awfmGpuSearchHits
 * (awfm_gpu.hip); nothing here synchronises with the host except a scratch (re)allocation.
 */
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <cstddef>

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rocprim/rocprim.hpp> /* awfmGpuSortHits: the list of hits sorted with its length known on the host */

#include "awfm_ordered_kernel.h"
#include "awfm_amino_lookup_kernel.h"
#include "awfm_count_order_kernel.h"

namespace {

inline size_t alignUp256(size_t v) { return (v + 255) / 256 * 256; }

/* the list of hits of a sparse search (awfmGpuSearchHitsCompact) before anything is appended: empty entries, count 0 */
__global__ void __launch_bounds__(256) fillSparseKernel(unsigned *__restrict__ kmers, ulonglong2 *__restrict__ ranges, unsigned cap,
                                                        unsigned *__restrict__ count) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i < cap) {
    kmers[i] = 0xFFFFFFFFu; /* sorts behind every k-mer */
    ranges[i] = make_ulonglong2(1ull, 0ull);
  }
  if (i == 0) *count = 0u;
}
/* every search path that lists its hits starts with this (the bucketed path's lookupPrepKernel does it beside its other work) */
inline hipError_t fillSparseList(const SparseOut *sparse, hipStream_t s) {
  if (!sparse || !sparse->count) return hipSuccess;
  hipLaunchKernelGGL(fillSparseKernel, dim3((sparse->cap + 255u) / 256u), dim3(256), 0, s, sparse->kmers, sparse->ranges, sparse->cap, sparse->count);
  return hipGetLastError();
}
constexpr size_t kOrderCounterBytes = 131072; /* 256 B for the count + 8 XCDs x kTicketGroups x 8 waves x 256 B of ticket counters */
static_assert(256 + 8 * kTicketGroups * 8 * 256 <= kOrderCounterBytes, "ticket counters");

template <class Kernel>
unsigned residentGrid(const AwFmGpuIndex *g, Kernel kernel, size_t dynamicLds = 0, int threads = kThreads) {
  int perCU = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kernel, threads, dynamicLds) != hipSuccess || perCU < 1) perCU = 4;
  if (perCU > 8) perCU = 8;
  return (unsigned)g->numCUs * (unsigned)perCU;
}

/* does the image step two characters per block read?  ($AWFM_GPU_PAIR=0 leaves the pair image out altogether) */
inline bool pairSteps(const AwFmGpuIndex *g) { return g->dev.pairBlocks != nullptr; }

template <bool NARROW, bool VARLEN, bool PAIR = false, bool TOUCH = false, bool BUCKET = false, bool LIST = false>
enum AwFmReturnCode launchOrderedKernel(AwFmGpuIndex *g, hipStream_t s, uint32_t len, unsigned depth, const ulonglong2 *table,
                                        unsigned long long nq, const void *recs, const unsigned *generalCount, ulonglong2 *rng,
                                        uint32_t *dCounts, const OrderTouch *touch = nullptr, const unsigned *bucketStart = nullptr,
                                        const BucketFormat bucketFmt = BucketFormat(), const SparseOut *sparse = nullptr,
                                        const unsigned *skipSampleAlive = nullptr, unsigned skipSamples = 0u) {
  constexpr int G = 4;
  /* dynamic LDS: the 32-bit superblock bases of the pair image (images below 2^32 positions) */
  const bool superInLds = PAIR && NARROW && awfmPairSuperInLds(g);
  const size_t lds = superInLds ? (size_t)g->dev.numPairSuper * 64u : 0u; /* the 16 pair bases of every superblock */
  DevIndex dev = g->dev;
  dev.pairSuperInLds = superInLds ? 1u : 0u;
  constexpr int threads = orderedThreads(PAIR);
  unsigned grid = residentGrid(g, orderedSearchKernel<G, NARROW, VARLEN, PAIR, TOUCH, BUCKET, LIST>, lds, threads);
  const unsigned long long blocks = (nq + threads / G - 1) / (threads / G);
  if (blocks < grid) grid = (unsigned)blocks;
  if (grid >= 8u) grid &= ~7u; /* a multiple of the 8 XCDs, so that every XCD gets the same number of workgroups */
  /* measurement hook (bench.py): HIP events around the kernel on its launch stream (events [2],[3]; when the call's
   * dominant kernel was the lookup kernel its own events [0],[1] are the ones reported) */
  /* (the events ride on the kernel's own dispatch -- hipExtLaunchKernelGGL -- instead of being recorded around it: a
   * recorded event is a packet of its own and left the queue idle for about 5 us each, 24 us per search) */
  const bool timed = g->orderTiming[2] != nullptr; /* this search has an entry in the timing log */
  OrderSkip<VARLEN> skip;
  if constexpr (VARLEN) {
    skip.sampleAlive = skipSampleAlive;
    skip.samples = skipSamples;
  }
  AWFM_LAUNCH_WITH_EVENTS((orderedSearchKernel<G, NARROW, VARLEN, PAIR, TOUCH, BUCKET, LIST>), dim3(grid ? grid : 1u), dim3(threads), (unsigned)lds, s,
                        timed ? g->orderTiming[2] : nullptr, timed ? g->orderTiming[3] : nullptr, dev, recs, nq, generalCount, len, depth, table,
                        rng, dCounts, (unsigned *)generalCount + 64, touch ? *touch : OrderTouch(), bucketStart, bucketFmt,
                        sparse ? *sparse : SparseOut(),
                        /* chunks a ticket is worth: 10^8 random 21-mers 3.92 (1), 3.49 (2), 3.47 (4), 3.50 (8), 3.65 ms (16); planted
                         * 21-mers (round 4, the batches this kernel still sees whole: the others end in lookupSearchKernel) 5.06 (1),
                         * 5.01 (2), 5.19 (3), 5.18 (4): the records in flight on an XCD span fewer buckets.  Mixed lengths: no gain */
                        BUCKET ? 2u : 1u, skip);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  g->orderTimedKernel = timed;
  if (timed) g->orderLog[(g->orderLogCount - 1u) % AwFmGpuIndex::kOrderLogMax].kernel = true;
  return AwFmSuccess;
}

/* the general kernel over what a hits-only front end left (the last *leftCount of the `numRecs` 8-byte or 16-byte records at
 * `recs`: k-mers with ambiguity characters, none or more than 32 characters, survivors beyond a round's slots) -- the last
 * kernel of a search: it carries the event that says the scratch slot is free again */
template <bool NARROW, bool CSR>
enum AwFmReturnCode launchLeftover(AwFmGpuIndex *g, hipStream_t s, const uint8_t *dChars, const unsigned long long *off, uint32_t len,
                                   unsigned long long nq, ulonglong2 *rng, uint32_t *dCounts, const void *recs, unsigned recordBytes,
                                   unsigned indexAt, const unsigned *leftCount, const SparseOut *sparse, bool last = true) {
  const unsigned grid = residentGrid(g, searchKernel<false, 4, CSR, false, NARROW, true>);
  AWFM_LAUNCH_WITH_EVENTS((searchKernel<false, 4, CSR, false, NARROW, true>), dim3(grid), dim3(kThreads), 0u, s, nullptr, last ? g->orderDoneEvent : nullptr,
                          g->dev, dChars, off, len, nq, rng, dCounts, (unsigned long long *)nullptr, (const unsigned char *)recs, recordBytes,
                          indexAt, nq, leftCount, sparse ? *sparse : SparseOut(), (const unsigned *)nullptr, 0u);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  if (last) g->orderDoneArmed = g->orderDoneEvent != nullptr;
  return AwFmSuccess;
}

/* the search over 16-byte records (partitionRecordsKernel), then the general kernel over the last bin */
template <bool NARROW, bool VARLEN>
enum AwFmReturnCode launchOrdered(AwFmGpuIndex *g, hipStream_t s, const uint8_t *dChars, const unsigned long long *off,
                                  uint32_t len, unsigned depth, const ulonglong2 *table, unsigned long long nq,
                                  const void *recs, const unsigned *generalCount, ulonglong2 *rng, uint32_t *dCounts,
                                  const OrderTouch *touch = nullptr, const SparseOut *sparse = nullptr,
                                  const unsigned *skipSampleAlive = nullptr, unsigned skipSamples = 0u) {
  enum AwFmReturnCode rc;
  const BucketFormat none = BucketFormat();
  if (touch) /* instrumented launch: the variant the image would run */
    rc = pairSteps(g) ? launchOrderedKernel<NARROW, VARLEN, true, true>(g, s, len, depth, table, nq, recs, generalCount, rng, dCounts, touch)
                      : launchOrderedKernel<NARROW, VARLEN, false, true>(g, s, len, depth, table, nq, recs, generalCount, rng, dCounts, touch);
  else
    rc = pairSteps(g) ? launchOrderedKernel<NARROW, VARLEN, true>(g, s, len, depth, table, nq, recs, generalCount, rng, dCounts, nullptr, nullptr, none, sparse, skipSampleAlive, skipSamples)
                      : launchOrderedKernel<NARROW, VARLEN, false>(g, s, len, depth, table, nq, recs, generalCount, rng, dCounts, nullptr, nullptr, none, sparse, skipSampleAlive, skipSamples);
  if (rc != AwFmSuccess) return rc;
  return launchLeftover<NARROW, VARLEN>(g, s, dChars, off, len, nq, rng, dCounts, recs, (unsigned)sizeof(QueryRec), (unsigned)offsetof(QueryRec, index),
                                        generalCount, sparse);
}

/* the search over bucketed 8-byte records (partitionKernel), then the general kernel over the last bucket */
template <bool NARROW>
enum AwFmReturnCode launchBucketed(AwFmGpuIndex *g, hipStream_t s, const uint8_t *dChars, uint32_t len, unsigned depth,
                                   const ulonglong2 *table, unsigned long long nq, const void *recs, const unsigned *bucketStart,
                                   const BucketFormat &fmt, const unsigned *generalCount, ulonglong2 *rng, uint32_t *dCounts,
                                   bool packed, const OrderTouch *touch, const SparseOut *sparse, bool countRecords = false) {
  /* countRecords: `sparse` names the array of {k-mer number, count} the ordered kernel fills in search order; the caller takes
   * them home and launches the general kernel over the last bucket itself (the last kernel of a search) */
  const bool pair = pairSteps(g);
  enum AwFmReturnCode rc;
  if (touch)
    rc = pair ? launchOrderedKernel<NARROW, false, true, true, true>(g, s, len, depth, table, nq, recs, generalCount, rng, dCounts, touch, bucketStart, fmt)
              : launchOrderedKernel<NARROW, false, false, true, true>(g, s, len, depth, table, nq, recs, generalCount, rng, dCounts, touch, bucketStart, fmt);
  else if (sparse && sparse->count) /* the list of hits: collected per wave */
    rc = pair ? launchOrderedKernel<NARROW, false, true, false, true, true>(g, s, len, depth, table, nq, recs, generalCount, rng, dCounts, nullptr, bucketStart, fmt, sparse)
              : launchOrderedKernel<NARROW, false, false, false, true, true>(g, s, len, depth, table, nq, recs, generalCount, rng, dCounts, nullptr, bucketStart, fmt, sparse);
  else
    rc = pair ? launchOrderedKernel<NARROW, false, true, false, true>(g, s, len, depth, table, nq, recs, generalCount, rng, dCounts, nullptr, bucketStart, fmt, sparse)
              : launchOrderedKernel<NARROW, false, false, false, true>(g, s, len, depth, table, nq, recs, generalCount, rng, dCounts, nullptr, bucketStart, fmt, sparse);
  if (rc != AwFmSuccess || packed || countRecords) return rc; /* bit-packed k-mers: every one of them is covered */
  /* the last bucket: k-mers with ambiguity characters; a record of it is the query number alone */
  return launchLeftover<NARROW, false>(g, s, dChars, nullptr, len, nq, rng, dCounts, recs, 8u, 0u, generalCount, sparse);
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
    if (const char *env = awfmKnob(AWFM_KNOB_ORDERED)) mode = atoi(env) != 0;
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
/* awfm_device.h.  Two passes over the finished table: the saturated lengths are counted (so that the side list is
 * allocated exactly and the in-place rewrite cannot fail half-way), then deepNextKernel rewrites the entries. */
int awfmGpuDeepSeedAddNext(AwFmGpuIndex *g, void *table, unsigned deepK, unsigned format, void **bigOut, unsigned *numBigOut) {
  *numBigOut = 0;
  if (!g || !table || format == 0u || deepK == 0) return 0;
  if (g->amino ? deepK > 7u : (!g->dev.pairBlocks || deepK > 16u)) return 0;
  if (awfmKnob(AWFM_KNOB_DEEP_NEXT) && atoi(awfmKnob(AWFM_KNOB_DEEP_NEXT)) == 0) return 0; /* comparison runs */
  DeviceGuard guard(g->device);
  if (format == 2u) { /* (awfmGpuBuildDeepSeedTable) the table is complete but for its bits; *bigOut holds the long lengths */
    unsigned *dCount = nullptr;
    if (hipMalloc((void **)&dCount, 16) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    unsigned numBig = 0;
    bool ok = awfmGpuSetupMemset(dCount, 0, 16) == hipSuccess;
    if (ok) {
      DevIndex dev = g->dev;
      dev.deepNarrow = 2u;
      dev.deepBigBySp = (const unsigned *)*bigOut;
      dev.pairSuperInLds = 0u;
      if (g->amino) {
        unsigned long long numEntries = 1;
        for (unsigned k = 0; k < deepK; k++) numEntries *= 20ull;
        const unsigned grid = residentGrid(g, aminoDeepNextKernel<false>, 0, kThreads);
        hipLaunchKernelGGL(aminoDeepNextKernel<false>, dim3(grid ? grid : 1u), dim3(kThreads), 0, awfmGpuSetupStream, dev, (uint2 *)table, numEntries, (unsigned *)nullptr, dCount);
      } else {
        constexpr int threads = orderedThreads(true);
        const unsigned grid = residentGrid(g, deepNextKernel<false>, 0, threads);
        hipLaunchKernelGGL(deepNextKernel<false>, dim3(grid ? grid : 1u), dim3(threads), 0, awfmGpuSetupStream, dev, (uint2 *)table, 1ull << (2u * deepK), (unsigned *)nullptr, dCount);
      }
      ok = hipGetLastError() == hipSuccess && awfmGpuSetupSync() == hipSuccess && awfmGpuSetupToHost(&numBig, dCount, 4) == hipSuccess;
    }
    (void)hipFree(dCount);
    if (!ok) {
      (void)hipGetLastError();
      awfmGpuSetError("deep seed table: the pass that adds the next-step bits failed");
      return -1;
    }
    *numBigOut = numBig;
    return 1;
  }
  *bigOut = nullptr;
  if (g->dev.bwtLength >= (1ull << 32)) return 0;
  if (g->amino) { /* {sp, length12 | next20 << 12}, the long lengths by sp >> 11 (awfm_device.h) */
    unsigned long long numEntries = 1;
    for (unsigned k = 0; k < deepK; k++) numEntries *= 20ull;
    const size_t bigWords = (size_t)(g->dev.bwtLength >> kAminoDeepBigShift) + 1u;
    unsigned *dBig = nullptr;
    if (hipMalloc((void **)&dBig, (bigWords + 4u) * 4u) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    unsigned numBig = 0;
    bool ok = awfmGpuSetupMemset(dBig, 0, (bigWords + 4u) * 4u) == hipSuccess;
    if (ok) {
      const unsigned grid = residentGrid(g, aminoDeepNextKernel<true>, 0, kThreads);
      hipLaunchKernelGGL(aminoDeepNextKernel<true>, dim3(grid ? grid : 1u), dim3(kThreads), 0, awfmGpuSetupStream, g->dev, (uint2 *)table, numEntries, dBig, dBig + bigWords);
      ok = hipGetLastError() == hipSuccess && awfmGpuSetupSync() == hipSuccess &&
           awfmGpuSetupToHost(&numBig, dBig + bigWords, 4) == hipSuccess;
    }
    if (!ok) {
      (void)hipGetLastError();
      (void)hipFree(dBig);
      awfmGpuSetError("deep seed table: the pass that adds the next-step bits failed");
      return -1;
    }
    *bigOut = dBig;
    *numBigOut = numBig;
    return 1;
  }
  const unsigned long long numEntries = 1ull << (2u * deepK);
  /* the lengths that do not fit the entries' 16 bits, by where their ranges begin (awfm_device.h: deepBigLength), allocated
   * before the in-place rewrite so that it cannot fail half-way; + the word that counts them */
  const size_t bigWords = (size_t)(g->dev.bwtLength >> kDeepBigShift) + 1u;
  unsigned *dBig = nullptr;
  if (hipMalloc((void **)&dBig, (bigWords + 4u) * 4u) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  bool ok = awfmGpuSetupMemset(dBig, 0, (bigWords + 4u) * 4u) == hipSuccess;
  unsigned numBig = 0;
  if (ok) {
    const bool superInLds = awfmPairSuperInLds(g);
    const size_t lds = superInLds ? (size_t)g->dev.numPairSuper * 64u : 0u;
    DevIndex dev = g->dev;
    dev.pairSuperInLds = superInLds ? 1u : 0u;
    constexpr int threads = orderedThreads(true);
    unsigned grid = residentGrid(g, deepNextKernel<true>, lds, threads);
    hipLaunchKernelGGL(deepNextKernel<true>, dim3(grid ? grid : 1u), dim3(threads), lds, awfmGpuSetupStream, dev, (uint2 *)table, numEntries, dBig, dBig + bigWords);
    ok = hipGetLastError() == hipSuccess && awfmGpuSetupSync() == hipSuccess &&
         awfmGpuSetupToHost(&numBig, dBig + bigWords, 4) == hipSuccess;
  }
  if (!ok) {
    /* the table may have been rewritten in part: the caller must not use it */
    (void)hipGetLastError();
    (void)hipFree(dBig);
    awfmGpuSetError("deep seed table: the pass that adds the next-step bits failed");
    return -1;
  }
  *bigOut = dBig;
  *numBigOut = numBig;
  return 1;
}

/* did the last bucketed search on the image keep its k-mers by encodeLookupKernel?  When the search left the decision to
 * the device (orderLookup == 2) the sample's count is read back here -- a reporting call, it waits for the device.  The
 * caller holds orderMutex. */
static bool lastSearchLookedUpFirst(AwFmGpuIndex *g) {
  if (g->orderLookup != 2) return g->orderLookup == 1;
  if (!g->orderSampleAt) return false;
  DeviceGuard guard(g->device);
  unsigned alive = 0;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&alive, g->orderSampleAt, sizeof alive, hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return alive * 4u < g->orderSamples;
}

/* 1 when the last seed-order search on the image kept its k-mers by encodeLookupKernel: that kernel is then the one
 * awfmGpuLastOrderedKernelMs timed (reporting; may wait for the device) */
extern "C" int awfmGpuLastOrderedKernelIsLookup(AwFmGpuIndex *g) {
  if (!g) return 0;
  std::lock_guard<std::mutex> lock(g->orderMutex);
  return lastSearchLookedUpFirst(g) ? 1 : 0;
}

/* k-mers the last bucketed seed-order search on the image ordered and searched: the batch, or what encodeLookupKernel kept
 * of it (reporting; waits for the device) */
extern "C" uint64_t awfmGpuLastOrderedKept(AwFmGpuIndex *g) {
  if (!g) return 0;
  std::lock_guard<std::mutex> lock(g->orderMutex); /* the word lives in the scratch a search may be re-allocating */
  if (!g->orderKeptAt) return 0;
  DeviceGuard guard(g->device);
  unsigned kept = 0;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&kept, g->orderKeptAt, sizeof kept, hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  if (g->orderLookupFused && g->orderFusedKeptAt && lastSearchLookedUpFirst(g)) {
    /* the survivors were searched by the kernel that looked them up: its waves counted them (what `kept` holds are the
     * k-mers left to the general kernel) */
    unsigned words[kFusedCounters * 16u];
    if (hipMemcpy(words, g->orderFusedKeptAt, sizeof words, hipMemcpyDeviceToHost) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    uint64_t total = kept;
    for (unsigned i = 0; i < kFusedCounters; i++) total += words[i * 16u];
    return total;
  }
  return kept;
}

/* milliseconds the dominant kernel of the last awfmGpuSearchHits* on this image took -- encodeLookupKernel when the batch
 * was one for "lookup first", orderedSearchKernel otherwise -- when the call ran with $AWFM_GPU_TIME_ORDERED set (waits
 * for it); negative when there is none */
extern "C" double awfmGpuLastOrderedKernelMs(AwFmGpuIndex *g) {
  if (!g) return -1.0;
  std::lock_guard<std::mutex> lock(g->orderMutex);
  const bool front = lastSearchLookedUpFirst(g);
  if (front ? !g->orderTimedFront : !g->orderTimedKernel) return -1.0;
  hipEvent_t from = g->orderTiming[front ? 0 : 2], to = g->orderTiming[front ? 1 : 3];
  if (!from || !to) return -1.0;
  float ms = 0.0f;
  if (hipEventSynchronize(to) != hipSuccess || hipEventElapsedTime(&ms, from, to) != hipSuccess) return -1.0;
  return (double)ms;
}

/* The brackets of every search since the log was last read (at most the last kOrderLogMax), oldest first: frontMs[i] =
 * encodeLookupKernel's (< 0: that search had none), kernelMs[i] = orderedSearchKernel's.  Waits for the last of them; the
 * log is empty afterwards.  Lets a caller time every step of a loop without a host wait inside the loop. */
extern "C" int awfmGpuOrderedKernelLog(AwFmGpuIndex *g, double *frontMs, double *kernelMs, int max) {
  if (!g || !frontMs || !kernelMs || max <= 0) return 0;
  std::lock_guard<std::mutex> lock(g->orderMutex);
  DeviceGuard guard(g->device);
  unsigned long long n = g->orderLogCount;
  if (n > AwFmGpuIndex::kOrderLogMax) n = AwFmGpuIndex::kOrderLogMax;
  if (n > (unsigned long long)max) n = (unsigned long long)max;
  for (unsigned long long i = 0; i < n; i++) {
    const AwFmGpuIndex::OrderLogEntry &e = g->orderLog[(g->orderLogCount - n + i) % AwFmGpuIndex::kOrderLogMax];
    float ms = 0.0f;
    frontMs[i] = e.front && hipEventSynchronize(e.ev[1]) == hipSuccess && hipEventElapsedTime(&ms, e.ev[0], e.ev[1]) == hipSuccess ? (double)ms : -1.0;
    kernelMs[i] = e.kernel && hipEventSynchronize(e.ev[3]) == hipSuccess && hipEventElapsedTime(&ms, e.ev[2], e.ev[3]) == hipSuccess ? (double)ms : -1.0;
  }
  (void)hipGetLastError();
  g->orderLogCount = 0;
  return (int)n;
}

/* the other of the two: orderedSearchKernel over the k-mers encodeLookupKernel kept (negative when the last search did
 * not look up first, or was not timed) */
extern "C" double awfmGpuLastOrderedSearchKernelMs(AwFmGpuIndex *g) {
  if (!g) return -1.0;
  std::lock_guard<std::mutex> lock(g->orderMutex);
  if (!g->orderTimedKernel || !g->orderTiming[2] || !g->orderTiming[3]) return -1.0;
  float ms = 0.0f;
  if (hipEventSynchronize(g->orderTiming[3]) != hipSuccess || hipEventElapsedTime(&ms, g->orderTiming[2], g->orderTiming[3]) != hipSuccess)
    return -1.0;
  return (double)ms;
}

/* see include/awfm_gpu.h */
extern "C" int awfmGpuLastLookupFront(AwFmGpuIndex *g) {
  if (!g) return -1;
  std::lock_guard<std::mutex> lock(g->orderMutex);
  const AwFmGpuIndex::LookupPredict &p = g->predict;
  return p.searches == 0u ? -1 : p.lastFront;
}

extern "C" int awfmGpuLastSearchWasExactLookup(AwFmGpuIndex *g) { return g ? g->lastSearchExact : 0; }

extern "C" int awfmGpuSearchHitsIsOrdered(const AwFmGpuIndex *g, int hasOffsets, uint32_t fixedLength, uint64_t numQueries) {
  if (!g) return 0;
  if (g->amino) return 0;
  if (!(g->kernel == AWFM_GPU_KERNEL_AUTO || g->kernel == AWFM_GPU_KERNEL_GROUP4)) return 0;
  return orderedApplies(g, hasOffsets != 0, fixedLength, numQueries, nullptr, nullptr) ? 1 : 0;
}

/* 1: the batch was searched; 0: the ordered path does not apply (caller runs the general kernel); <0: -AwFmReturnCode */
static int orderedSearch(AwFmGpuIndex *g, hipStream_t s, const uint8_t *dChars, const unsigned long long *off,
                         uint32_t fixedLength, unsigned long long nq, ulonglong2 *rng, uint32_t *dCounts, bool packed,
                         bool rangesOfHitsOnly, const OrderTouch *touch, uint64_t *recordBytesOut,
                         const SparseOut *sparse);

/* what the seed-order search returns when its scratch cannot be allocated: distinguishable from "does not apply" (0), so
 * that a caller with another way to the same results (the general kernel needs no scratch) takes it and the others say
 * what really happened */
constexpr int kOrderNoScratch = -(int)AwFmAllocationFailure;

/* ---- gates: who used a piece of scratch last ---- */
/* `s` is about to use what the gate guards: it waits for the last user unless that was the same stream (in the same thread:
 * hipStreamPerThread names a different stream in every thread), which is ordered behind it already */
/* A recorded event -- whether by hipEventRecord or as the stop event of a kernel launch -- leaves the queue idle for about
 * 5 us (measured: the launches behind one start 4.5-7 us late), and a search that is followed by another on the SAME
 * stream, the usual case, needs none.  So for a stream the caller created (a handle that names the same stream from every
 * thread) nothing is recorded when the use ends: the event is recorded on that stream when a user on ANOTHER stream shows
 * up, behind whatever the first stream has been given since -- more than needed, never less.  (Such a stream must therefore
 * outlive its searches on the image: destroy it only after synchronising it.)  The null stream and hipStreamPerThread,
 * whose handles do not name one stream, record eagerly. */
static bool gateLazy(hipStream_t s) { return s != nullptr && s != hipStreamPerThread && s != hipStreamLegacy; }
static hipError_t gateEnter(AwFmGpuIndex::StreamGate &gate, hipStream_t s) {
  if (!gate.done) {
    const hipError_t e = hipEventCreate(&gate.done);
    if (e != hipSuccess) return e;
  }
  if (!gate.recorded && !gate.pending) return hipSuccess;
  if (gate.lastStream == s && s != nullptr && (gateLazy(s) || gate.lastThread == std::this_thread::get_id())) return hipSuccess;
  if (gate.lastStream == s && s == nullptr && gate.lastThread == std::this_thread::get_id()) return hipSuccess;
  if (gate.pending) { /* the last user left no event behind: now one is needed */
    /* (a stream that was synchronised and destroyed without awfmGpuStreamRetire: its handle no longer answers a query, and
     * everything it was given is over) */
    const hipError_t alive = hipStreamQuery(gate.lastStream);
    if ((alive != hipSuccess && alive != hipErrorNotReady) || hipEventRecord(gate.done, gate.lastStream) != hipSuccess) {
      (void)hipGetLastError();
      const hipError_t e = hipDeviceSynchronize(); /* that stream is gone: whatever it was given has to be over */
      gate.pending = gate.recorded = false;
      return e;
    }
    gate.pending = false;
    gate.recorded = true;
  }
  return hipStreamWaitEvent(s, gate.done, 0);
}
/* `s` has enqueued its last use; armed: the last kernel was launched with gate.done as its stop event (hipExtLaunchKernelGGL),
 * which costs nothing, where a hipEventRecord is a packet of its own and about 5 us of idle queue */
static hipError_t gateLeave(AwFmGpuIndex::StreamGate &gate, hipStream_t s, bool armed) {
  if (gateLazy(s)) {
    gate.pending = true;
    gate.recorded = false;
  } else {
    if (!armed) {
      const hipError_t e = hipEventRecord(gate.done, s);
      if (e != hipSuccess) return e;
    }
    gate.recorded = true;
    gate.pending = false;
  }
  gate.lastStream = s;
  gate.lastThread = std::this_thread::get_id();
  return hipSuccess;
}

/* see include/awfm_gpu.h: the caller is about to destroy `stream`.  Whatever the image remembers of it -- a gate whose event
 * was left to be recorded on that stream when another stream shows up -- is settled now, while the handle is still good:
 * the event is recorded (behind everything the stream has been given), and the stream's handle is forgotten. */
extern "C" void awfmGpuStreamRetire(AwFmGpuIndex *g, void *stream) {
  if (!g) return;
  hipStream_t s = (hipStream_t)stream;
  DeviceGuard guard(g->device);
  std::lock_guard<std::mutex> lock(g->orderMutex);
  auto settle = [&](AwFmGpuIndex::StreamGate &gate) {
    if (gate.lastStream != s || !(gate.pending || gate.recorded)) return;
    if (gate.pending) {
      if (!gate.done && hipEventCreate(&gate.done) != hipSuccess) gate.done = nullptr;
      if (gate.done && hipEventRecord(gate.done, s) == hipSuccess) {
        gate.recorded = true;
      } else { /* no event to be had: whatever the stream was given has to be over before the handle goes */
        (void)hipGetLastError();
        (void)hipStreamSynchronize(s);
        gate.recorded = false;
      }
      gate.pending = false;
    }
    gate.lastStream = nullptr; /* a new stream that gets the same handle is another stream: it waits for the event */
    gate.lastThread = std::thread::id();
  };
  for (auto &slot : g->orderSlot) settle(slot.gate);
  settle(g->sparseGate);
}

/* the scratch slot of a search on stream `s` (the caller holds orderMutex): the one this stream used last, else the one
 * that has rested longest; `s` waits for that slot's last user */
static hipError_t orderBeginSlot(AwFmGpuIndex *g, hipStream_t s) {
  int pick = -1;
  for (int i = 0; i < AwFmGpuIndex::kOrderSlots; i++) {
    const AwFmGpuIndex::StreamGate &gate = g->orderSlot[i].gate;
    if ((gate.recorded || gate.pending) && gate.lastStream == s && (gateLazy(s) || gate.lastThread == std::this_thread::get_id())) pick = i;
  }
  if (pick < 0) {
    pick = 0;
    for (int i = 1; i < AwFmGpuIndex::kOrderSlots; i++)
      if (g->orderSlot[i].lastUse < g->orderSlot[pick].lastUse) pick = i;
  }
  AwFmGpuIndex::OrderSlot &slot = g->orderSlot[pick];
  slot.lastUse = ++g->orderUses;
  g->orderPrevParity = slot.prepParity; /* (a search that keeps the two sample words in step sets it again) */
  slot.prepParity = -1;
  g->orderCur = pick;
  g->dOrder = slot.mem;
  g->orderBytes = slot.bytes;
  g->orderDoneArmed = false;
  const hipError_t e = gateEnter(slot.gate, s);
  g->orderDoneEvent = e == hipSuccess && !gateLazy(s) ? slot.gate.done : nullptr;
  return e;
}
static hipError_t orderEndSlot(AwFmGpuIndex *g, hipStream_t s) {
  const hipError_t e = gateLeave(g->orderSlot[g->orderCur].gate, s, g->orderDoneArmed);
  g->orderDoneEvent = nullptr;
  g->orderDoneArmed = false;
  return e;
}

/* Every exit of a search behind orderBeginSlot ends the slot's use -- the failing ones too (advisor, round 4: a kernel launch
 * that failed half-way left the kernels already enqueued on the slot's scratch unrecorded in its gate, and the next search on
 * another stream would not have waited for them). */
struct OrderSlotScope {
  AwFmGpuIndex *g;
  hipStream_t s;
  bool ended = false;
  OrderSlotScope(AwFmGpuIndex *image, hipStream_t stream) : g(image), s(stream) {}
  OrderSlotScope(const OrderSlotScope &) = delete;
  OrderSlotScope &operator=(const OrderSlotScope &) = delete;
  hipError_t end() {
    ended = true;
    return orderEndSlot(g, s);
  }
  ~OrderSlotScope() {
    if (!ended) {
      (void)orderEndSlot(g, s);
      (void)hipGetLastError();
    }
  }
};

/* scratch of the current slot, grown when needed; the caller holds orderMutex.  false: no memory (the general kernel needs none) */
static bool ensureOrderScratch(AwFmGpuIndex *g, size_t bytes) {
  AwFmGpuIndex::OrderSlot &slot = g->orderSlot[g->orderCur];
  if (bytes <= slot.bytes) return true;
  /* hipFree waits for every stream of the device, so nothing still reads the old scratch */
  if (slot.mem) (void)hipFree(slot.mem);
  slot.mem = nullptr;
  slot.bytes = 0;
  g->dOrder = nullptr;
  g->orderBytes = 0;
  g->orderKeptAt = nullptr;
  g->orderSampleAt = nullptr;
  g->orderPrevParity = -1;
  const size_t want = bytes + bytes / 8;
  if (hipMalloc(&slot.mem, want) != hipSuccess) {
    (void)hipGetLastError();
    setError("seed-order search: no device memory for its scratch");
    return false;
  }
  slot.bytes = want;
  g->dOrder = slot.mem;
  g->orderBytes = want;
  return true;
}

/* encodeCodes4Kernel<K> for the batch's k-mer length */
template <unsigned K>
static void launchEncode4At(unsigned len, unsigned grid, size_t lds, hipStream_t s, const uint8_t *dChars, const BucketFormat &fmt,
                            unsigned long long nq, unsigned long long *codes, unsigned *hist, unsigned binsPad,
                            const unsigned *sampleAlive, unsigned samples) {
  if (len == K)
    hipLaunchKernelGGL((encodeCodes4Kernel<K>), dim3(grid), dim3(256), lds, s, dChars, fmt, nq, codes, hist, binsPad, sampleAlive, samples);
  else if constexpr (K > 1u)
    launchEncode4At<K - 1u>(len, grid, lds, s, dChars, fmt, nq, codes, hist, binsPad, sampleAlive, samples);
}
static void launchEncode4(unsigned len, unsigned grid, size_t lds, hipStream_t s, const uint8_t *dChars, const BucketFormat &fmt,
                          unsigned long long nq, unsigned long long *codes, unsigned *hist, unsigned binsPad,
                          const unsigned *sampleAlive = nullptr, unsigned samples = 0u) {
  launchEncode4At<32u>(len, grid, lds, s, dChars, fmt, nq, codes, hist, binsPad, sampleAlive, samples);
}

/* lookupSearchKernel<K, NARROW> for the batch's k-mer length */
template <unsigned K, bool NARROW>
static void launchLookupSearchAt(unsigned len, unsigned grid, size_t lds, hipStream_t s, const DevIndex &dev, const uint8_t *dChars,
                                 const BucketFormat &fmt, unsigned useNext, unsigned long long nq, unsigned long long *codes,
                                 unsigned *numbers, unsigned *shareCount, unsigned *hist, unsigned binsPad,
                                 const unsigned *sampleAlive, unsigned samples, ulonglong2 *rng, unsigned *dCounts,
                                 const SparseOut &sparse, unsigned *keptCounters, hipEvent_t start, hipEvent_t stop) {
  if (len == K)
    AWFM_LAUNCH_WITH_EVENTS((lookupSearchKernel<K, NARROW>), dim3(grid), dim3(256), (unsigned)lds, s, start, stop, dev, dChars, fmt, useNext, nq, codes,
                          numbers, shareCount, hist, binsPad, sampleAlive, samples, rng, dCounts, sparse, keptCounters);
  else if constexpr (K > 1u)
    launchLookupSearchAt<K - 1u, NARROW>(len, grid, lds, s, dev, dChars, fmt, useNext, nq, codes, numbers, shareCount, hist, binsPad, sampleAlive, samples,
                                         rng, dCounts, sparse, keptCounters, start, stop);
}

/* ---- lookup prediction (AwFmGpuIndex::LookupPredict) ---- */
enum { kFrontBoth = 0, kFrontLookupOnly = 1, kFrontOrderedOnly = 2 };
constexpr unsigned kPredictHoldoff = 8, kPredictSamples = 16384, kPredictNumberMask = (1u << 22) - 1u;
/* which front end(s) a sampled search of fixed-length k-mers launches, from the newest verdict that has reached the host
 * (nothing waits for one); the caller holds orderMutex.  $AWFM_GPU_LOOKUP_PREDICT=0: always both (round 4). */
static int predictFront(AwFmGpuIndex *g, unsigned fixedLength, unsigned chooseOf = kPredictSamples) {
  AwFmGpuIndex::LookupPredict &p = g->predict;
  if (const char *env = awfmKnob(AWFM_KNOB_LOOKUP_PREDICT))
    if (atoi(env) == 0) return kFrontBoth;
  if (!p.verdictHost) return kFrontBoth;
  /* the newest verdict: {tag of its search, k-mers of its sample alive}; the tag says what the host needs to know about that
   * search -- its number, the front end(s) it launched, its k-mer length -- however long ago it was enqueued */
  const unsigned long long v = *(volatile unsigned long long *)p.verdictHost;
  const unsigned tag = (unsigned)(v >> 32), alive = (unsigned)v;
  const unsigned number = tag & kPredictNumberMask, front = (tag >> 22) & 3u, length = (tag >> 24) & 63u;
  if (number == 0u) return kFrontBoth;
  /* a verdict is judged -- and followed -- only by a call of its own kind (the tag's k-mer length; 0: a mixed-length batch),
   * whose threshold `chooseOf` is the one its search was launched under: another kind of batch on the same image neither
   * counts as a miss nor resets the agreement (advisor, round 5) */
  if (length != fixedLength) return p.holdoff ? (p.holdoff--, kFrontBoth) : kFrontBoth;
  const bool lookup = alive * 4u < chooseOf; /* lookupChosen's rule (chooseOf: the sample's size, or more of it where the lookup kernel pays up to a higher share) */
  if (number != p.lastJudged) {
    p.lastJudged = number;
    if ((front == kFrontLookupOnly && !lookup) || (front == kFrontOrderedOnly && lookup)) {
      /* a stream whose batches keep changing character: every miss doubles the searches that launch both front ends */
      p.holdoff = p.holdoffNext;
      p.holdoffNext = p.holdoffNext < 1024u ? 2u * p.holdoffNext : 1024u;
      p.agreed = 0;
    } else if (front != kFrontBoth && ++p.agreed >= 64u) {
      p.holdoffNext = kPredictHoldoff;
    }
  }
  if (p.holdoff) {
    p.holdoff--;
    return kFrontBoth;
  }
  return lookup ? kFrontLookupOnly : kFrontOrderedOnly;
}
/* the tag of the next sampled search (never 0 in its number: "no verdict yet") */
static unsigned predictTag(AwFmGpuIndex *g, int front, unsigned fixedLength) {
  AwFmGpuIndex::LookupPredict &p = g->predict;
  p.searches = (p.searches + 1u) & kPredictNumberMask;
  if (p.searches == 0u) p.searches = 1u;
  p.lastFront = front;
  return p.searches | ((unsigned)front << 22) | ((fixedLength & 63u) << 24);
}

/* fillNoHitKernel -> encodeCodes4Kernel -> bucketScanSharesKernel -> partitionKernel -> orderedSearchKernel<BUCKET> ->
 * searchKernel<INDIRECT> on the caller's stream; the caller holds orderMutex.  Return values as awfmGpuOrderedSearch. */
static int bucketedSearch(AwFmGpuIndex *g, hipStream_t s, const uint8_t *dChars, uint32_t fixedLength, unsigned depth,
                          const ulonglong2 *table, unsigned long long nq, ulonglong2 *rng, uint32_t *dCounts, bool packed,
                          bool rangesOfHitsOnly, const OrderTouch *touch, const BucketFormat &fmt, const SparseOut *sparse) {
  const unsigned bins = (1u << fmt.bucketBits) + 1u, binsPad = (bins + 3u) & ~3u;
  /* [counters 128 KB][hist -> sub-run starts: 8 shares x binsPad][cursors: 8 x binsPad][bucketStart: bins + 1]
   * [codes: nq x 8 unless packed][records: nq x 8] */
  const size_t histAt = kOrderCounterBytes, cursorsAt = histAt + alignUp256((size_t)kShares * binsPad * 4u);
  const size_t startAt = cursorsAt + alignUp256((size_t)kShares * binsPad * 4u);
  const size_t codesAt = startAt + alignUp256((bins + 1u) * 4u), recsAt = codesAt + (packed ? 0u : alignUp256(nq * 8u));
  /* "lookup first" (encodeLookupKernel): ASCII k-mers that start from the narrow deeper table, results dense or as the list of
   * hits (results in search order owe an entry to every k-mer).  Its k-mer numbers: 4 bytes per k-mer behind the records;
   * share counts and the sample's count: in the counter block, beyond the ticket counters. */
  const bool narrow = awfmImageNarrow(g);
  const bool lookupCapable = !packed && !touch && table == g->dev.deepSeed && g->dev.deepNarrow != 0u &&
                             !(sparse && sparse->kmers && !sparse->count);
  const char *lookupEnv = awfmKnob(AWFM_KNOB_LOOKUP_FIRST); /* 0: never, 1: whenever it applies; unset: by a sample of the batch */
  const bool lookupWanted = lookupCapable && (lookupEnv ? atoi(lookupEnv) != 0 : nq >= (1ull << 20));
  const size_t numbersAt = recsAt + alignUp256(nq * 8u);
  /* counts only, dense (awfm_count_order_kernel.h): the ordered kernel's {number, count} records in search order, and the
   * same records by the leading bits of their numbers */
  const unsigned countBuckets = countOrderBuckets(nq);
  const bool countRecords = !sparse && !touch && dCounts && !rng && countBuckets != 0u;
  constexpr unsigned countShift = kCountShift;
  const size_t countInAt = numbersAt + (lookupWanted ? alignUp256(nq * 4u) : 0u);
  const size_t countOutAt = countInAt + (countRecords ? alignUp256(nq * 8u) : 0u);
  const size_t total = countOutAt + (countRecords ? alignUp256(((size_t)countBuckets << countShift) * 8u) : 0u);
  if (orderBeginSlot(g, s) != hipSuccess) {
    setError("seed-order search: could not order the use of its scratch across streams");
    return -(int)AwFmGeneralFailure;
  }
  OrderSlotScope slotScope(g, s);
  if (!ensureOrderScratch(g, total)) return kOrderNoScratch;
  constexpr size_t kShareCountAt = 98304, kSampleAt = kShareCountAt + kShares * kShareCountStride * 4u; /* bytes into the counter block (tickets end at 65792) */
  constexpr size_t kKeptAt = 102400; /* lookupSearchKernel's survivor counters: kFusedCounters words a line apart */
  constexpr size_t kCountCursorsAt = 66048; /* countScatterKernel's cursors (zeroed with the counters) */
  static_assert(256 + 8 * kTicketGroups * 8 * 256 <= kCountCursorsAt && kCountCursorsAt + kCountBucketsMax * 4u <= kShareCountAt, "counter block");
  static_assert(256 + 8 * kTicketGroups * 8 * 256 <= kShareCountAt && kSampleAt + 4 <= kKeptAt && kKeptAt + kFusedCounters * 64u <= kOrderCounterBytes, "counter block");
#define BUCKET_TRY(call)                    \
  do {                                      \
    hipError_t e__ = (call);                \
    if (e__ != hipSuccess) {                \
      setError(#call, e__);                 \
      return -(int)AwFmGeneralFailure;      \
    }                                       \
  } while (0)
  uint8_t *w = (uint8_t *)g->dOrder;
  unsigned *generalCount = (unsigned *)w;
  unsigned *hist = (unsigned *)(w + histAt), *cursors = (unsigned *)(w + cursorsAt), *bucketStart = (unsigned *)(w + startAt);
  const unsigned long long *codes = packed ? (const unsigned long long *)dChars : (const unsigned long long *)(w + codesAt);
  unsigned long long *recs = (unsigned long long *)(w + recsAt);
  /* grids: multiples of the 8 shares (workgroup b works on share b % 8) */
  const unsigned long long perShare256 = (shareSize(nq) + 255ull) / 256ull;
  unsigned encodeGrid = (unsigned)(perShare256 * kShares < (unsigned long long)g->numCUs * 8u ? perShare256 * kShares : (unsigned long long)g->numCUs * 8u);
  encodeGrid = (encodeGrid + kShares - 1u) / kShares * kShares;
  unsigned *shareCount = (unsigned *)(w + kShareCountAt), *numbers = (unsigned *)(w + numbersAt);
  const unsigned useNext = g->dev.deepNext != 0u && pairSteps(g) && fixedLength >= depth + 2u ? 1u : 0u;
  /* lookup first: forced ($AWFM_GPU_LOOKUP_FIRST=1), or left to a sample of the batch -- 16384 k-mers at a fixed stride; the
   * pass pays when fewer than a quarter of them are alive after the table.  The sample's count stays on the device: both
   * front ends are launched and the one it does not choose returns at once (lookupChosen), so the search never waits
   * for the host -- unless the sample of an earlier search of this k-mer length has reached the host by now:
   * then only the front end it chose is launched (predictFront). */
  constexpr unsigned kSamples = 16384;
  const bool forced = lookupWanted && lookupEnv && atoi(lookupEnv) == 1;
  bool bySample = lookupWanted && !forced;
  /* the scratch's counters zeroed, the list's fill and the sample in one launch (lookupPrepKernel); a sampled batch has 2^20
   * k-mers and more */
  const bool prepFused = bySample && nq >= kSamples;
  AwFmGpuIndex::LookupPredict &predict = g->predict;
  if (prepFused && !predict.verdictHost) {
    if (hipHostMalloc((void **)&predict.verdictHost, 64, hipHostMallocDefault) == hipSuccess) {
      memset(predict.verdictHost, 0, 64);
    } else {
      (void)hipGetLastError();
      predict.verdictHost = nullptr; /* no verdicts: every search launches both front ends */
    }
  }
  const int front = prepFused ? predictFront(g, fixedLength) : kFrontBoth;
  const unsigned *sampleAlive = nullptr;
  bool lookupFirst = forced;
  if (prepFused) {
    /* the two sample words, a line each (kSampleAt, kSampleAt + 128): this search adds to the one the last search on the
     * slot left zero and zeroes the other */
    int parity = g->orderPrevParity;
    if (parity < 0) {
      BUCKET_TRY(hipMemsetAsync(w + kSampleAt, 0, 256, s));
      parity = 0;
    }
    unsigned long long *aliveOut = (unsigned long long *)(w + kSampleAt + 128u * (unsigned)parity);
    unsigned long long *aliveNext = (unsigned long long *)(w + kSampleAt + 128u * (unsigned)(1 - parity));
    /* what the launch zeroes: everything the memset did but the sample words -- or, when only the lookup kernel follows, the
     * line of the general kernel's count and the survivor counters */
    uint4 *zeroA = (uint4 *)w, *zeroB = (uint4 *)(w + kSampleAt + 256u);
    unsigned vecsA = (unsigned)(kSampleAt / 16u), vecsB = (unsigned)((startAt - kSampleAt - 256u) / 16u);
    if (front == kFrontLookupOnly) {
      vecsA = 256u / 16u;
      zeroB = (uint4 *)(w + kKeptAt);
      vecsB = kFusedCounters * 64u / 16u;
    }
    static_assert(kSamples == kPredictSamples, "the verdict is judged against the sample's size");
    const unsigned number = predictTag(g, front, fixedLength);
    const unsigned prepGrid = front == kFrontLookupOnly && !(sparse && sparse->count) ? kSamples / 256u : 2u * (kSamples / 256u);
    hipLaunchKernelGGL(lookupPrepKernel<false>, dim3(prepGrid), dim3(256), 0, s, g->dev, dChars, fixedLength, depth, useNext, nq, kSamples, aliveOut,
                       aliveNext, zeroA, vecsA, zeroB, vecsB, sparse && sparse->count ? *sparse : SparseOut(), predict.verdictHost, number);
    BUCKET_TRY(hipGetLastError());
    g->orderSlot[g->orderCur].prepParity = 1 - parity;
    if (!sparse) {
      hipLaunchKernelGGL(fillNoHitKernel, dim3((unsigned)g->numCUs * 8u), dim3(256), 0, s,
                         rangesOfHitsOnly && dCounts ? (ulonglong2 *)nullptr : rng, dCounts, nq);
      BUCKET_TRY(hipGetLastError());
    }
    g->orderSampleAt = (const unsigned *)aliveOut; /* (the count is the word's low half) */
    if (front == kFrontBoth) sampleAlive = (const unsigned *)aliveOut;
    else {
      bySample = false; /* the host has decided: one front end, which does not look at the sample */
      lookupFirst = front == kFrontLookupOnly;
    }
  } else {
    BUCKET_TRY(hipMemsetAsync(w, 0, startAt, s)); /* the count, the ticket counters, the histograms, the cursors */
    BUCKET_TRY(fillSparseList(sparse, s)); /* (lookupPrepKernel fills the list beside its other work) */
    if (!sparse) { /* a sparse search lists its hits: there is nothing to pre-fill */
      hipLaunchKernelGGL(fillNoHitKernel, dim3((unsigned)g->numCUs * 8u), dim3(256), 0, s,
                         rangesOfHitsOnly && dCounts ? (ulonglong2 *)nullptr : rng, dCounts, nq);
      BUCKET_TRY(hipGetLastError());
    }
    bySample = false; /* (a batch of fewer k-mers than the sample takes: forced, or not looked up first at all) */
    g->orderSampleAt = (const unsigned *)(w + kSampleAt);
  }
  const bool lookupOnly = prepFused && front == kFrontLookupOnly; /* no ordering passes, no ordered kernel behind the lookup kernel */
  g->orderLookup = bySample ? 2 : (lookupFirst ? 1 : 0);
  g->orderSamples = kSamples;
  g->orderKeptAt = lookupOnly ? generalCount : bucketStart + bins; /* the k-mers left to the general kernel / the scan's total */
  g->orderTimedFront = false;
  if (lookupFirst || bySample) {
    const bool timed = g->orderTiming[0] != nullptr; /* this search has an entry in the timing log: the events ride on the dispatch */
    /* the k-mers still alive after the table are searched by the kernel that looked them up (lookupSearchKernel) */
    g->orderLookupFused = true;
    g->orderFusedKeptAt = (const unsigned *)(w + kKeptAt);
    {
      const bool pairOff = !pairSteps(g);
      const bool superInLds = !pairOff && narrow && awfmPairSuperInLds(g);
      DevIndex dev = g->dev;
      dev.pairSuperInLds = superInLds ? 1u : 0u;
      const size_t lds = superInLds ? (size_t)g->dev.numPairSuper * 64u : 0u;
      /* persistent grid: what is resident (7 workgroups per CU; the 64-bit instantiation: 80 registers, 6), a multiple of the 8 shares */
      const unsigned perCU = narrow ? 7u : 6u;
      unsigned fusedGrid = (unsigned)(perShare256 * kShares < (unsigned long long)g->numCUs * perCU ? perShare256 * kShares : (unsigned long long)g->numCUs * perCU);
      fusedGrid = (fusedGrid + kShares - 1u) / kShares * kShares;
      /* A workgroup takes its share 1024 k-mers a trip.  A small batch is a few trips per workgroup -- 6.8 for the 1.25 * 10^7
       * k-mers of a shard of an 8-GPU run on 1792 workgroups: most take 7, and the chip idles while they finish -- so the grid
       * is trimmed to the workgroups that share the trips evenly (1744 x 7) */
      {
        const unsigned long long tripsPerShare = (shareSize(nq) + 1023ull) / 1024ull, groupsPerShare = fusedGrid / kShares;
        if (groupsPerShare > 0u && tripsPerShare > groupsPerShare) {
          const unsigned long long tripsEach = (tripsPerShare + groupsPerShare - 1ull) / groupsPerShare;
          if (tripsEach <= 64ull) fusedGrid = (unsigned)((tripsPerShare + tripsEach - 1ull) / tripsEach) * kShares;
        }
      }
      /* (lookup only: what the kernel does not search itself goes to the END of the record array, 8 bytes a k-mer number,
       * counted in the general kernel's word -- no code words, no numbers, no histogram) */
#define AWFM_LOOKUP_GO(NR)                                                                                                               \
  launchLookupSearchAt<32u, NR>(fixedLength, fusedGrid, lds, s, dev, dChars, fmt, useNext | (pairOff ? 2u : 0u) | (lookupOnly ? 4u : 0u), nq, \
                                (unsigned long long *)(w + codesAt), lookupOnly ? (unsigned *)recs : numbers, lookupOnly ? generalCount : shareCount, \
                                hist, binsPad, sampleAlive, kSamples, rng, dCounts, sparse ? *sparse : SparseOut(),                      \
                                (unsigned *)(w + kKeptAt), timed ? g->orderTiming[0] : nullptr, timed ? g->orderTiming[1] : nullptr)
      if (narrow) AWFM_LOOKUP_GO(true);
      else AWFM_LOOKUP_GO(false);
#undef AWFM_LOOKUP_GO
    }
    g->orderTimedFront = timed;
    if (timed) g->orderLog[(g->orderLogCount - 1u) % AwFmGpuIndex::kOrderLogMax].front = true;
    BUCKET_TRY(hipGetLastError());
  }
  if (lookupOnly) {
    /* what the lookup kernel left (k-mers with ambiguity characters, survivors beyond a round's slots): the general kernel
     * over the tail of the record array -- the last kernel of the search: it carries the event that says the slot is free */
    enum AwFmReturnCode rc = AwFmSuccess;
    if (!packed)
      rc = narrow ? launchLeftover<true, false>(g, s, dChars, nullptr, fixedLength, nq, rng, dCounts, recs, 8u, 0u, generalCount, sparse)
                  : launchLeftover<false, false>(g, s, dChars, nullptr, fixedLength, nq, rng, dCounts, recs, 8u, 0u, generalCount, sparse);
    if (rc != AwFmSuccess) return -(int)rc;
    BUCKET_TRY(slotScope.end());
    return 1;
  }
  if (lookupFirst) {
    /* (forced or decided by the host: the other front end is not launched) */
  } else if (packed)
    hipLaunchKernelGGL((encodeCodesKernel<true>), dim3(encodeGrid), dim3(256), bins * 4u, s, dChars, fixedLength, fmt, nq,
                       (unsigned long long *)nullptr, hist, binsPad);
  else
    launchEncode4(fixedLength, encodeGrid, bins * 4u, s, dChars, fmt, nq, (unsigned long long *)(w + codesAt), hist, binsPad, sampleAlive, kSamples);
  BUCKET_TRY(hipGetLastError());
  hipLaunchKernelGGL(bucketScanSharesKernel, dim3(1), dim3(1024), 0, s, hist, bins, binsPad, bucketStart, generalCount, (unsigned)nq);
  BUCKET_TRY(hipGetLastError());
  const size_t partitionLds = (size_t)kPartitionTile * 8u + 3u * binsPad * 4u;
  static std::once_flag ldsOnce;
  static hipError_t ldsError = hipSuccess;
  std::call_once(ldsOnce, [] {
    ldsError = hipFuncSetAttribute((const void *)partitionKernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(kPartitionTile * 8u + 3u * (((1u << kBucketBitsMax) + 4u) & ~3u) * 4u));
  });
  BUCKET_TRY(ldsError);
  const unsigned long long tilesPerShare = shareSize(nq) / kPartitionTile;
  unsigned partitionGrid = (unsigned)(tilesPerShare * kShares < (unsigned long long)g->numCUs ? tilesPerShare * kShares : (unsigned long long)g->numCUs);
  partitionGrid = (partitionGrid + kShares - 1u) / kShares * kShares;
  hipLaunchKernelGGL(partitionKernel, dim3(partitionGrid), dim3(kPartitionThreads), partitionLds, s, codes, fixedLength, fmt, nq,
                     (const unsigned *)hist, cursors, recs, packed ? 0u : 1u,
                     lookupFirst || bySample ? (const unsigned *)shareCount : (const unsigned *)nullptr,
                     lookupFirst || bySample ? (const unsigned *)numbers : (const unsigned *)nullptr, sampleAlive, kSamples);
  BUCKET_TRY(hipGetLastError());
  const SparseOut countOut{nullptr, 0u, (unsigned *)(w + countInAt), nullptr};
  const SparseOut *results = countRecords ? &countOut : sparse;
  const enum AwFmReturnCode rc =
      narrow ? launchBucketed<true>(g, s, dChars, fixedLength, depth, table, nq, recs, bucketStart, fmt, generalCount, rng, dCounts, packed, touch, results, countRecords)
                         : launchBucketed<false>(g, s, dChars, fixedLength, depth, table, nq, recs, bucketStart, fmt, generalCount, rng, dCounts, packed, touch, results, countRecords);
  if (rc != AwFmSuccess) return -(int)rc;
  if (countRecords) { /* the counts home from search order (the number of records: where the general kernel's bucket begins) */
    const unsigned *ordered = bucketStart + (1u << fmt.bucketBits);
    unsigned *countCursors = (unsigned *)(w + kCountCursorsAt);
    const unsigned long long tiles = (nq + kCountScatterTile - 1ull) / kCountScatterTile;
    const unsigned scatterGrid = (unsigned)(tiles < (unsigned long long)g->numCUs * 8u ? tiles : (unsigned long long)g->numCUs * 8u); /* (16 KB of LDS each) */
    hipLaunchKernelGGL(countScatterKernel, dim3(scatterGrid ? scatterGrid : 1u), dim3(kCountScatterThreads), 0, s, (const uint2 *)(w + countInAt), ordered,
                       countBuckets, (uint2 *)(w + countOutAt), countCursors);
    BUCKET_TRY(hipGetLastError());
    static std::once_flag placeOnce;
    static hipError_t placeError = hipSuccess;
    std::call_once(placeOnce, [] {
      placeError = hipFuncSetAttribute((const void *)countPlaceKernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((1u << kCountShift) * 4u));
    });
    BUCKET_TRY(placeError);
    hipLaunchKernelGGL(countPlaceKernel, dim3(countBuckets), dim3(kCountPlaceThreads), (size_t)(1u << kCountShift) * 4u, s, (const uint2 *)(w + countOutAt),
                       (const unsigned *)countCursors, nq, dCounts);
    BUCKET_TRY(hipGetLastError());
    if (!packed) { /* the k-mers the order does not hold: stored at counts[number] */
      const enum AwFmReturnCode left = narrow ? launchLeftover<true, false>(g, s, dChars, nullptr, fixedLength, nq, rng, dCounts, recs, 8u, 0u, generalCount, nullptr)
                                              : launchLeftover<false, false>(g, s, dChars, nullptr, fixedLength, nq, rng, dCounts, recs, 8u, 0u, generalCount, nullptr);
      if (left != AwFmSuccess) return -(int)left;
    }
  }
  BUCKET_TRY(slotScope.end());
#undef BUCKET_TRY
  return 1;
}

int awfmGpuOrderedSearch(AwFmGpuIndex *g, hipStream_t s, const uint8_t *dChars, const unsigned long long *off,
                         uint32_t fixedLength, unsigned long long nq, ulonglong2 *rng, uint32_t *dCounts, bool packed,
                         bool rangesOfHitsOnly) {
  return orderedSearch(g, s, dChars, off, fixedLength, nq, rng, dCounts, packed, rangesOfHitsOnly, nullptr, nullptr, nullptr);
}

/* the tables of the k-mer lengths below the deeper table's, built by the first search that can use them (on the primary
 * image, for its lanes too); nullptr: there are none (no memory, or the image is not one for them) */
static const uint2 *ensureLengthTables(AwFmGpuIndex *g) {
  AwFmGpuIndex *p = g->shares ? g->shares : g;
  const unsigned need = g->dev.deepK - 1u;
  std::lock_guard<std::mutex> lock(p->lengthMutex);
  if (p->dLengthTable && p->lengthDepths >= need) return (const uint2 *)p->dLengthTable;
  /* after a failed attempt the next ones wait: every 64th call tries again (memory may have been freed since) */
  if (p->lengthTried && (++p->lengthRetryIn & 63u) != 0u) return nullptr;
  p->lengthTried = true;
  { /* the same headroom rule as the other automatic tables: three times the table free on the device */
    const uint64_t tableBytes = awfmLengthTableAt(need + 1u) * 8ull;
    size_t freeBytes = 0, totalBytes = 0;
    if (hipMemGetInfo(&freeBytes, &totalBytes) != hipSuccess || (uint64_t)freeBytes / 3u < tableBytes) {
      (void)hipGetLastError();
      return nullptr;
    }
  }
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  void *table = nullptr;
  uint64_t bytes = 0;
  void *big = nullptr;
  if (!awfmGpuBuildLengthTables(p, need, &table, &bytes, &big)) {
    (void)hipGetLastError();
    return nullptr;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (p->dLengthBig) (void)hipFree(p->dLengthBig);
  p->dLengthBig = big; /* (before the table: whoever finds the table under this mutex finds its long lengths) */
  p->dLengthTable = table;
  p->lengthDepths = need;
  p->lengthTableBytes = bytes;
  p->lengthTableBuildSeconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  if (awfmKnob(AWFM_KNOB_VERBOSE))
    fprintf(stderr, "[awfm length tables] lengths 1..%u: %.2f GB in %.3f s\n", need, (double)bytes * 1e-9, p->lengthTableBuildSeconds);
  return (const uint2 *)table;
}

/* encodeRecordsKernel -> bucketScanKernel -> partitionRecordsKernel -> orderedSearchKernel (16-byte records) ->
 * searchKernel<INDIRECT>; the caller holds orderMutex.  Return values as awfmGpuOrderedSearch.
 * Mixed-length batches on an image with the narrow deeper table: "lookup first" through one table per k-mer length
 * (awfm_mixed_lookup_kernel.h) -- forced ($AWFM_GPU_MIXED_LOOKUP=1), never (=0), or (unset, batches of >= 2^20 k-mers) left to
 * a sample of the batch that stays on the device: both front ends are launched, the one it does not choose returns at once. */
static int wideBucketedSearch(AwFmGpuIndex *g, hipStream_t s, const uint8_t *dChars, const unsigned long long *off, uint32_t fixedLength,
                              unsigned depth, const ulonglong2 *table, unsigned long long nq, ulonglong2 *rng, uint32_t *dCounts,
                              bool rangesOfHitsOnly, const OrderTouch *touch, const SparseOut *sparse, bool lookupAlways = false) {
  constexpr unsigned bins = (1u << kBucketBitsMax) + 1u, binsPad = (bins + 3u) & ~3u;
  const size_t histAt = kOrderCounterBytes, cursorsAt = histAt + alignUp256(bins * 4u), startAt = cursorsAt + alignUp256(bins * 4u);
  /* (lookupAlways: a batch below the size from which ordering pays -- the lookup kernel or nothing: no records) */
  const size_t recordBytes = lookupAlways ? 0u : alignUp256(nq * sizeof(QueryRec));
  const size_t inAt = startAt + alignUp256((bins + 1u) * 4u), outAt = inAt + recordBytes;
  const size_t leftAt = outAt + recordBytes;
  /* the sample: 16384 k-mers at a fixed stride; the lookup kernel is chosen when fewer than THREE quarters of them are still
   * alive after their table entry (lookupChosen compares 4 x alive with the number it is given: 3 x the sample's size) --
   * the records of the other path cost a mixed-length batch 2.5 ms per 10^8 before anything is searched, and its search
   * kernel is the slower one even when half the k-mers survive (8..30-mers, half drawn from the text: 6.4 against 9.8 ms;
   * 18..30-mers: 9.1 against 10.7 ms) */
  constexpr unsigned kSamples = 16384, kChooseOf = 3u * kSamples;
  const char *mixedEnv = awfmKnob(AWFM_KNOB_MIXED_LOOKUP);
  const bool mixedCapable = off && !touch && !g->amino && g->dev.deepSeed && g->dev.deepNarrow != 0u &&
                            g->dev.deepK >= 2u && g->dev.deepK <= 16u && g->dev.seedK < g->dev.deepK &&
                            !(sparse && sparse->kmers && !sparse->count) /* results in search order owe an entry to every k-mer */;
  const bool mixedForced = mixedCapable && (lookupAlways || (mixedEnv && atoi(mixedEnv) == 1));
  const bool mixedWanted = mixedCapable && (mixedEnv ? atoi(mixedEnv) != 0 : nq >= (1ull << 20)) && (mixedForced || nq >= kSamples);
  const uint2 *lengthTable = mixedWanted ? ensureLengthTables(g) : nullptr;
  const bool bySample = lengthTable && !mixedForced;
  if (lookupAlways && !lengthTable) return 0; /* (no tables after all: the general kernel) */
  /* round 5, as for fixed lengths (predictFront; a mixed-length batch has length 0 in its tag): the sample is taken on the
   * device as before and its verdict published to the host, and a batch whose predecessors' verdicts have arrived and agree
   * launches ONE front end -- the lookup kernel, which then takes whatever the batch is (correct for any batch), or the
   * 16-byte-record path alone */
  AwFmGpuIndex::LookupPredict &predict = g->predict;
  if (bySample && !predict.verdictHost) {
    if (hipHostMalloc((void **)&predict.verdictHost, 64, hipHostMallocDefault) == hipSuccess) {
      memset(predict.verdictHost, 0, 64);
    } else {
      (void)hipGetLastError();
      predict.verdictHost = nullptr;
    }
  }
  const int front = bySample ? predictFront(g, 0u, kChooseOf) : kFrontBoth;
  const bool lookupOnly = lengthTable && (mixedForced || front == kFrontLookupOnly);
  const bool lookupRuns = lengthTable && front != kFrontOrderedOnly;
  /* counts only, dense, and nothing but the lookup kernel: it stores every k-mer's count itself, a round's at a time in whole
   * lines (bit 4 of useNext) -- no pre-fill, no 4-byte stores at k-mer numbers */
  /* (not for awfmGpuSearchHitsSparse, whose point is that the ranges of the k-mers WITHOUT hits are not written: 16 of the 20
   * bytes per k-mer; advisor, round 5) */
  const bool wholeCounts = lookupOnly && !sparse && (dCounts || rng) && !(rangesOfHitsOnly && dCounts && rng);
  const size_t total = leftAt + (lengthTable ? alignUp256(nq * 8u) : 0u);
  /* in the counter block, beyond the ticket counters (which end at 65792): the leftover count, the sample's count, the
   * survivor counters (kFusedCounters words a line apart) */
  constexpr size_t kLeftCountAt = 98304, kSampleAt = 98304 + 256, kKeptAt = 102400;
  static_assert(kKeptAt + kFusedCounters * 64u <= kOrderCounterBytes, "counter block");
  if (orderBeginSlot(g, s) != hipSuccess) {
    setError("seed-order search: could not order the use of its scratch across streams");
    return -(int)AwFmGeneralFailure;
  }
  OrderSlotScope slotScope(g, s);
  if (!ensureOrderScratch(g, total)) return kOrderNoScratch;
#define WIDE_TRY(call)                      \
  do {                                      \
    hipError_t e__ = (call);                \
    if (e__ != hipSuccess) {                \
      setError(#call, e__);                 \
      return -(int)AwFmGeneralFailure;      \
    }                                       \
  } while (0)
  uint8_t *w = (uint8_t *)g->dOrder;
  unsigned *generalCount = (unsigned *)w;
  unsigned *hist = (unsigned *)(w + histAt), *cursors = (unsigned *)(w + cursorsAt), *bucketStart = (unsigned *)(w + startAt);
  QueryRec *recsIn = (QueryRec *)(w + inAt), *recsOut = (QueryRec *)(w + outAt);
  WIDE_TRY(hipMemsetAsync(w, 0, startAt, s));
  WIDE_TRY(fillSparseList(sparse, s));
  if (!sparse && !wholeCounts) {
    hipLaunchKernelGGL(fillNoHitKernel, dim3((unsigned)g->numCUs * 8u), dim3(256), 0, s,
                       rangesOfHitsOnly && dCounts ? (ulonglong2 *)nullptr : rng, dCounts, nq);
    WIDE_TRY(hipGetLastError());
  }
  const unsigned *sampleAlive = nullptr;
  if (bySample) { /* the sample: always taken (the next searches' prediction), consulted on the device when both front ends run */
    const bool pairOff = !pairSteps(g);
    const unsigned useNext = (g->dev.deepNext != 0u && !pairOff ? 1u : 0u) | (pairOff ? 2u : 0u);
    unsigned *sampleWord = (unsigned *)(w + kSampleAt);
    static_assert(kSamples == kPredictSamples, "the verdict is judged against the sample's size");
    const unsigned number = predictTag(g, front, 0u);
    WIDE_TRY(awfmGpuLaunchMixedSample(g, s, lengthTable, dChars, off, nq, useNext, kSamples, (unsigned long long *)sampleWord, predict.verdictHost, number));
    if (front == kFrontBoth) sampleAlive = sampleWord;
  }
  if (lookupRuns) {
    const bool pairOff = !pairSteps(g);
    unsigned useNext = (g->dev.deepNext != 0u && !pairOff ? 1u : 0u) | (pairOff ? 2u : 0u);
    if (wholeCounts) useNext |= 16u;
    /* the superblock bases of the pair image are read from memory: 24 KB of them in LDS (a 3.1 Gbp image) would leave room
     * for 3 workgroups per CU where the survivors' slots alone allow 6 (10^8 8..30-mers: 6.36 against 6.74 ms) */
    const bool superInLds = false;
    unsigned *leftoverCount = (unsigned *)(w + kLeftCountAt), *sampleWord = (unsigned *)(w + kSampleAt), *kept = (unsigned *)(w + kKeptAt);
    unsigned long long *leftover = (unsigned long long *)(w + leftAt);
    g->orderLookup = sampleAlive ? 2 : 1;
    g->orderSampleAt = sampleWord;
    g->orderSamples = kChooseOf;
    g->orderLookupFused = true;
    g->orderFusedKeptAt = kept;
    g->orderKeptAt = leftoverCount;
    const SparseOut out = sparse ? *sparse : SparseOut();
    const bool timed = g->orderTiming[0] != nullptr; /* this search has an entry in the timing log: the events ride on the dispatch */
    WIDE_TRY(awfmGpuLaunchMixedLookup(g, s, timed ? g->orderTiming[0] : nullptr, timed ? g->orderTiming[1] : nullptr, lengthTable, dChars, off, nq,
                                      useNext, superInLds, sampleAlive, kChooseOf, rng, dCounts, out.count, out.cap, out.kmers, out.ranges, leftover,
                                      leftoverCount, kept));
    g->orderTimedFront = timed;
    if (timed) g->orderLog[(g->orderLogCount - 1u) % AwFmGpuIndex::kOrderLogMax].front = true;
    /* what the lookup kernel left: the last *leftoverCount records of the list */
    {
      const enum AwFmReturnCode left =
          awfmImageNarrow(g) ? launchLeftover<true, true>(g, s, dChars, off, fixedLength, nq, rng, dCounts, leftover, 8u, 0u, leftoverCount, sparse, lookupOnly)
                             : launchLeftover<false, true>(g, s, dChars, off, fixedLength, nq, rng, dCounts, leftover, 8u, 0u, leftoverCount, sparse, lookupOnly);
      if (left != AwFmSuccess) return -(int)left; /* (not lookup only: more kernels follow, and the last one carries the slot's event) */
    }
    if (lookupOnly) { /* forced or predicted: the other front end is not launched */
      WIDE_TRY(slotScope.end());
      return 1;
    }
  } else if (bySample) { /* predicted: the 16-byte-record path alone */
    g->orderLookup = 0;
  }
  const unsigned long long encodeTiles = (nq + 255ull) / 256ull;
  const unsigned encodeGrid = (unsigned)(encodeTiles < (unsigned long long)g->numCUs * 8u ? encodeTiles : (unsigned long long)g->numCUs * 8u);
  const unsigned seedK = g->dev.seedK, deepK = g->dev.deepK;
  if (off)
    hipLaunchKernelGGL((encodeRecordsKernel<true>), dim3(encodeGrid), dim3(256), 0, s, dChars, off, fixedLength, depth, seedK, deepK, nq,
                       recsIn, hist, sampleAlive, kChooseOf);
  else
    hipLaunchKernelGGL((encodeRecordsKernel<false>), dim3(encodeGrid), dim3(256), 0, s, dChars, off, fixedLength, depth, seedK, deepK, nq,
                       recsIn, hist);
  WIDE_TRY(hipGetLastError());
  hipLaunchKernelGGL(bucketScanKernel, dim3(1), dim3(1024), 0, s, (const unsigned *)hist, bins, bucketStart, generalCount);
  WIDE_TRY(hipGetLastError());
  const size_t lds = (size_t)kWideTile * 16u + 3u * binsPad * 4u;
  static std::once_flag ldsOnce;
  static hipError_t ldsError = hipSuccess;
  std::call_once(ldsOnce, [] {
    ldsError = hipFuncSetAttribute((const void *)partitionRecordsKernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ldsError == hipSuccess)
      ldsError = hipFuncSetAttribute((const void *)partitionRecordsKernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  WIDE_TRY(ldsError);
  const unsigned long long tiles = (nq + kWideTile - 1ull) / kWideTile;
  const unsigned grid = (unsigned)(tiles < (unsigned long long)g->numCUs ? tiles : (unsigned long long)g->numCUs);
  if (off)
    hipLaunchKernelGGL((partitionRecordsKernel<true>), dim3(grid), dim3(kPartitionThreads), lds, s, (const QueryRec *)recsIn, depth, seedK,
                       deepK, nq, (const unsigned *)bucketStart, cursors, recsOut, sampleAlive, kChooseOf);
  else
    hipLaunchKernelGGL((partitionRecordsKernel<false>), dim3(grid), dim3(kPartitionThreads), lds, s, (const QueryRec *)recsIn, depth, seedK,
                       deepK, nq, (const unsigned *)bucketStart, cursors, recsOut);
  WIDE_TRY(hipGetLastError());
  const bool narrow = awfmImageNarrow(g);
  enum AwFmReturnCode rc;
#define WIDE_GO(NR, VL) \
  launchOrdered<NR, VL>(g, s, dChars, off, fixedLength, depth, table, nq, recsOut, generalCount, rng, dCounts, touch, sparse, sampleAlive, kChooseOf)
  if (off) rc = narrow ? WIDE_GO(true, true) : WIDE_GO(false, true);
  else rc = narrow ? WIDE_GO(true, false) : WIDE_GO(false, false);
#undef WIDE_GO
  if (rc != AwFmSuccess) return -(int)rc;
  WIDE_TRY(slotScope.end());
#undef WIDE_TRY
  return 1;
}

static int orderedSearch(AwFmGpuIndex *g, hipStream_t s, const uint8_t *dChars, const unsigned long long *off,
                         uint32_t fixedLength, unsigned long long nq, ulonglong2 *rng, uint32_t *dCounts, bool packed,
                         bool rangesOfHitsOnly, const OrderTouch *touch, uint64_t *recordBytesOut, const SparseOut *sparse) {
  unsigned depth = 0;
  const ulonglong2 *table = nullptr;
  if (packed && off) return 0;
  bool lookupAlways = false;
  if (!orderedApplies(g, off != nullptr, fixedLength, nq, &depth, &table)) {
    /* Mixed-length batches below the size from which ordering pays (2^23 k-mers) on an image that has, or can have, its
     * tables per k-mer length: mixedLookupSearchKernel reads what the general kernel reads of the deeper table and of the
     * blocks -- 256 entries per wave at once instead of 16 chains -- and one entry where that kernel walks a short k-mer
     * up from a letter range.  From 2^20 k-mers (below that the general kernel's single launch wins);
     * $AWFM_GPU_MIXED_LOOKUP=0 and a seed-order path that is switched off keep the general kernel. */
    int mode = g->orderMode;
    if (mode < 0)
      if (const char *env = awfmKnob(AWFM_KNOB_ORDERED)) mode = atoi(env) != 0;
    const char *mixedEnv = awfmKnob(AWFM_KNOB_MIXED_LOOKUP);
    lookupAlways = off && !touch && mode != 0 && !g->amino && nq < 0xFFFFFFFFull && g->dev.deepSeed &&
                   g->dev.deepNarrow != 0u && g->dev.deepK >= 2u && g->dev.deepK <= 16u && g->dev.seedK < g->dev.deepK &&
                   !(sparse && sparse->kmers && !sparse->count) && (mixedEnv ? atoi(mixedEnv) != 0 : nq >= (1ull << 20));
    if (!lookupAlways) return 0;
  }

  std::lock_guard<std::mutex> lock(g->orderMutex);
  g->orderLookup = 0;
  g->orderTimedFront = false;
  g->orderTimedKernel = false;
  for (int i = 0; i < 4; i++) g->orderTiming[i] = nullptr;
  if (awfmKnob(AWFM_KNOB_TIME_ORDERED)) { /* measurement hook: this search's entry of the timing log */
    DeviceGuard guard(g->device);
    const unsigned at = (unsigned)(g->orderLogCount % AwFmGpuIndex::kOrderLogMax);
    if (at >= g->orderLog.size()) g->orderLog.resize(at + 1u);
    AwFmGpuIndex::OrderLogEntry &entry = g->orderLog[at];
    bool ok = true;
    for (int i = 0; i < 4 && ok; i++)
      if (!entry.ev[i]) ok = hipEventCreate(&entry.ev[i]) == hipSuccess;
    if (ok) {
      entry.front = entry.kernel = false;
      for (int i = 0; i < 4; i++) g->orderTiming[i] = entry.ev[i];
      g->orderLogCount++;
    } else {
      (void)hipGetLastError();
    }
  }
  if (lookupAlways) {
    if (recordBytesOut) *recordBytesOut = 0u;
    return wideBucketedSearch(g, s, dChars, off, fixedLength, 0u, nullptr, nq, rng, dCounts, rangesOfHitsOnly, nullptr, sparse, true);
  }
  /* fixed-length batches whose records fit 8 bytes: counted and partitioned by the kernels of awfm_ordered_kernel.h */
  const BucketFormat bucketFmt = bucketFormat(depth, nq);
  if (!off && bucketFits(fixedLength, bucketFmt)) {
    if (recordBytesOut) *recordBytesOut = 8u;
    return bucketedSearch(g, s, dChars, fixedLength, depth, table, nq, rng, dCounts, packed, rangesOfHitsOnly, touch, bucketFmt, sparse);
  }
  /* everything else the path covers -- mixed-length batches, fixed-length k-mers whose record does not fit 8 bytes --
   * through the same two passes over 16-byte records (bit-packed k-mers of more than 24 characters: the caller unpacks them) */
  if (packed) return 0;
  if (recordBytesOut) *recordBytesOut = sizeof(QueryRec);
  return wideBucketedSearch(g, s, dChars, off, fixedLength, depth, table, nq, rng, dCounts, rangesOfHitsOnly, touch, sparse);
}


/* ------------------------------------------------------------------ amino: lookup first (awfm_amino_lookup_kernel.h) */

template <unsigned K, bool NARROW>
static void launchAminoLookupAt(unsigned len, unsigned grid, hipStream_t s, const DevIndex &dev, const uint8_t *dChars, unsigned long long nq,
                                const unsigned *sampleAlive, unsigned samples, ulonglong2 *rng, unsigned *dCounts, const SparseOut &sparse,
                                unsigned long long *leftover, unsigned *leftoverCount, unsigned *kept) {
  if (len == K)
    hipLaunchKernelGGL((aminoLookupSearchKernel<K, NARROW>), dim3(grid), dim3(256), 0, s, dev, dChars, nq, sampleAlive, samples, rng, dCounts, sparse,
                       leftover, leftoverCount, kept);
  else if constexpr (K > 2u)
    launchAminoLookupAt<K - 1u, NARROW>(len, grid, s, dev, dChars, nq, sampleAlive, samples, rng, dCounts, sparse, leftover, leftoverCount, kept);
}

/* Hits-only search of a large fixed-length amino batch through the device-only deeper table: 1 = searched, 0 = does not
 * apply (the caller runs the general kernel), < 0 = -AwFmReturnCode.  $AWFM_GPU_AMINO_LOOKUP=0|1: never / whenever it can
 * (no sample); unset: batches of >= 2^20 k-mers, by a sample of the batch. */
static int aminoLookupSearch(AwFmGpuIndex *g, hipStream_t s, const uint8_t *dChars, uint32_t fixedLength, unsigned long long nq,
                             ulonglong2 *rng, uint32_t *dCounts, bool rangesOfHitsOnly, const SparseOut *sparse) {
  constexpr unsigned kMaxLength = 19; /* the table's 7 characters + 12 in front of them (5 bits each in one word) */
  const bool narrow = awfmImageNarrow(g);
  if (!g->amino || g->dev.deepK == 0u || g->dev.deepNarrow != (narrow ? 1u : 2u) || g->dev.deepSeed == nullptr) return 0;
  if (fixedLength < g->dev.deepK || fixedLength > kMaxLength || fixedLength - g->dev.deepK > 12u || nq >= 0xFFFFFFFFull) return 0;
  if (!(g->kernel == AWFM_GPU_KERNEL_AUTO || g->kernel == AWFM_GPU_KERNEL_GROUP2)) return 0;
  const char *env = awfmKnob(AWFM_KNOB_AMINO_LOOKUP);
  if (env ? atoi(env) == 0 : nq < (1ull << 20)) return 0;
  const bool forced = env && atoi(env) == 1;
  constexpr unsigned kSamples = 16384;
  if (!forced && nq < kSamples) return 0;
  std::lock_guard<std::mutex> lock(g->orderMutex);
  /* [counters 4 KB: sample @0, leftover count @64 B, survivor counters @256 B][leftover list: nq x 8 B] */
  constexpr size_t kCounterBytes = 256u + kFusedCounters * 64u;
  const size_t listAt = alignUp256(kCounterBytes), total = listAt + alignUp256(nq * 8u);
#define AMINO_TRY(call)                     \
  do {                                      \
    hipError_t e__ = (call);                \
    if (e__ != hipSuccess) {                \
      setError(#call, e__);                 \
      return -(int)AwFmGeneralFailure;      \
    }                                       \
  } while (0)
  AMINO_TRY(orderBeginSlot(g, s));
  OrderSlotScope slotScope(g, s);
  if (!ensureOrderScratch(g, total)) return kOrderNoScratch;
  uint8_t *w = (uint8_t *)g->dOrder;
  unsigned *leftoverCount = (unsigned *)(w + 64), *kept = (unsigned *)(w + 256);
  unsigned long long *leftover = (unsigned long long *)(w + listAt);
  /* round 5, as the nucleotide bucketed search: the counters zeroed, the list pre-filled and the sample taken by ONE launch
   * (lookupPrepKernel<true>; the sample's two words at bytes 0 and 128 of the counter block), and only the kernel an earlier
   * search's sample named when its verdict has reached the host (predictFront: 1 = this kernel, 2 = the general kernel) */
  const bool prepFused = !forced;
  AwFmGpuIndex::LookupPredict &predict = g->predict;
  if (prepFused && !predict.verdictHost) {
    if (hipHostMalloc((void **)&predict.verdictHost, 64, hipHostMallocDefault) == hipSuccess) {
      memset(predict.verdictHost, 0, 64);
    } else {
      (void)hipGetLastError();
      predict.verdictHost = nullptr;
    }
  }
  /* the lookup kernel has a slot for every k-mer of a round (round 5): it is chosen while fewer than `percent` of the sample
   * are still alive after their entry (lookupChosen compares 4 x alive with the number it is given) -- the general kernel
   * keeps twice the chains in flight per wave, which wins when nearly every k-mer goes on for several steps (k-mers drawn
   * from the text) */
  constexpr unsigned percent = 90u;
  const unsigned chooseOf = (unsigned)((unsigned long long)kSamples * 4ull * percent / 100ull);
  const int front = prepFused ? predictFront(g, fixedLength, chooseOf) : kFrontBoth;
  const unsigned *sampleAlive = nullptr;
  unsigned *sampleWord = (unsigned *)w;
  if (prepFused) {
    int parity = g->orderPrevParity;
    if (parity < 0) {
      AMINO_TRY(hipMemsetAsync(w, 0, 256, s));
      parity = 0;
    }
    unsigned long long *aliveOut = (unsigned long long *)(w + 128u * (unsigned)parity);
    unsigned long long *aliveNext = (unsigned long long *)(w + 128u * (unsigned)(1 - parity));
    static_assert(kSamples == kPredictSamples, "the verdict is judged against the sample's size");
    const unsigned number = predictTag(g, front, fixedLength);
    hipLaunchKernelGGL(lookupPrepKernel<true>, dim3(sparse && sparse->count ? 2u * (kSamples / 256u) : kSamples / 256u), dim3(256), 0, s, g->dev, dChars,
                       fixedLength, g->dev.deepK, 0u, nq, kSamples, aliveOut, aliveNext, (uint4 *)(w + 64), 64u / 16u, (uint4 *)(w + 256),
                       (unsigned)(kFusedCounters * 64u / 16u), sparse && sparse->count ? *sparse : SparseOut(), predict.verdictHost, number);
    AMINO_TRY(hipGetLastError());
    g->orderSlot[g->orderCur].prepParity = 1 - parity;
    sampleWord = (unsigned *)aliveOut;
    if (front == kFrontBoth) sampleAlive = sampleWord;
  } else { /* forced: no sample */
    AMINO_TRY(hipMemsetAsync(w, 0, kCounterBytes, s));
    AMINO_TRY(fillSparseList(sparse, s));
  }
  if (!sparse) {
    hipLaunchKernelGGL(fillNoHitKernel, dim3((unsigned)g->numCUs * 8u), dim3(256), 0, s,
                       rangesOfHitsOnly && dCounts ? (ulonglong2 *)nullptr : rng, dCounts, nq);
    AMINO_TRY(hipGetLastError());
  }
  const bool lookupRuns = !(prepFused && front == kFrontOrderedOnly), generalRuns = !forced && !(prepFused && front == kFrontLookupOnly);
  g->orderLookup = sampleAlive ? 2 : (lookupRuns ? 1 : 0);
  g->orderSampleAt = sampleWord;
  g->orderSamples = chooseOf;
  g->orderLookupFused = true;
  g->orderFusedKeptAt = kept;
  g->orderKeptAt = leftoverCount;
  const SparseOut out = sparse ? *sparse : SparseOut();
  if (lookupRuns) {
    const unsigned long long rounds = (nq + 1023ull) / 1024ull; /* a workgroup takes 1024 k-mers a round */
    unsigned grid = narrow ? residentGrid(g, aminoLookupSearchKernel<10u, true>) : residentGrid(g, aminoLookupSearchKernel<10u, false>);
    if (rounds < grid) grid = (unsigned)rounds;
    if (narrow) launchAminoLookupAt<kMaxLength, true>(fixedLength, grid ? grid : 1u, s, g->dev, dChars, nq, sampleAlive, chooseOf, rng, dCounts, out, leftover, leftoverCount, kept);
    else launchAminoLookupAt<kMaxLength, false>(fixedLength, grid ? grid : 1u, s, g->dev, dChars, nq, sampleAlive, chooseOf, rng, dCounts, out, leftover, leftoverCount, kept);
    AMINO_TRY(hipGetLastError());
  }
  if (generalRuns) { /* the whole batch through the general kernel when the sample says so (it returns at once otherwise) */
#define AMINO_GENERAL(NR, ...)                                                                                                     \
  do {                                                                                                                             \
    const unsigned grid__ = residentGrid(g, searchKernel<true, 2, false, false, NR, __VA_ARGS__>);                                 \
    hipLaunchKernelGGL((searchKernel<true, 2, false, false, NR, __VA_ARGS__>), dim3(grid__), dim3(kThreads), 0, s, g->dev, dChars, \
                       (const unsigned long long *)nullptr, fixedLength, nq, rng, dCounts, (unsigned long long *)nullptr, AMINO_GENERAL_ARGS); \
  } while (0)
#define AMINO_GENERAL_ARGS (const unsigned char *)nullptr, 0u, 0u, 0ull, (const unsigned *)nullptr, out, sampleAlive, chooseOf
    if (narrow) AMINO_GENERAL(true, false);
    else AMINO_GENERAL(false, false);
#undef AMINO_GENERAL_ARGS
    AMINO_TRY(hipGetLastError());
  }
  if (lookupRuns) { /* what the lookup kernel left: the last *leftoverCount records of the list */
#define AMINO_GENERAL_ARGS (const unsigned char *)leftover, 8u, 0u, nq, (const unsigned *)leftoverCount, out, (const unsigned *)nullptr, 0u
    if (narrow) AMINO_GENERAL(true, true);
    else AMINO_GENERAL(false, true);
#undef AMINO_GENERAL_ARGS
#undef AMINO_GENERAL
    AMINO_TRY(hipGetLastError());
  }
  AMINO_TRY(slotScope.end());
#undef AMINO_TRY
  return 1;
}
int awfmGpuAminoLookupSearch(AwFmGpuIndex *g, hipStream_t s, const uint8_t *dChars, uint32_t fixedLength, unsigned long long nq,
                             ulonglong2 *rng, uint32_t *dCounts, bool rangesOfHitsOnly) {
  return aminoLookupSearch(g, s, dChars, fixedLength, nq, rng, dCounts, rangesOfHitsOnly, nullptr);
}

/* ------------------------------------------------------------------ awfmGpuSearch: exact ranges through the tables */

/* awfm_device.h.  Batches of 2^18 k-mers and more on a nucleotide image below 2^32 positions with its narrow deeper table
 * ($AWFM_GPU_EXACT_LOOKUP=0|1: never / whenever the image can): exactLookupSearchKernel (awfm_exact_lookup_kernel.h), then the
 * letter-by-letter general kernel over what it left (ambiguity characters, no or more than 32 characters).  K-mers shorter
 * than the deeper table need the tables per k-mer length: used when the image has them; built for a CSR batch of 2^20
 * k-mers and more as the hits-only search of such a batch would; without them a fixed-length batch of short k-mers is the
 * general kernel's, and the short k-mers of a CSR batch go to it through the list. */
int awfmGpuExactLookupSearch(AwFmGpuIndex *g, hipStream_t s, const uint8_t *dChars, const unsigned long long *off, uint32_t fixedLength,
                             unsigned long long nq, ulonglong2 *rng, uint32_t *dCounts) {
  if (g->amino || !g->dev.deepSeed || g->dev.deepNarrow == 0u || g->dev.deepK < 2u || g->dev.deepK > 16u ||
      g->dev.seedK >= g->dev.deepK || nq >= 0xFFFFFFFFull)
    return 0;
  if (!(g->kernel == AWFM_GPU_KERNEL_AUTO || g->kernel == AWFM_GPU_KERNEL_GROUP4)) return 0;
  if (!off && (fixedLength == 0u || fixedLength > 32u)) return 0;
  const char *env = awfmKnob(AWFM_KNOB_EXACT_LOOKUP);
  if (env ? atoi(env) == 0 : nq < (1ull << 18)) return 0;
  const bool forced = env && atoi(env) == 1;
  const uint2 *lengthTable = nullptr;
  if (off || fixedLength < g->dev.deepK) {
    AwFmGpuIndex *p = g->shares ? g->shares : g;
    {
      std::lock_guard<std::mutex> lock(p->lengthMutex);
      if (p->dLengthTable && p->lengthDepths >= g->dev.deepK - 1u) lengthTable = (const uint2 *)p->dLengthTable;
    }
    const char *mixedEnv = awfmKnob(AWFM_KNOB_MIXED_LOOKUP);
    /* (awfmGpuSearch allocates nothing by itself: the tables are used when a hits-only mixed-length batch has built them,
     * or built here when asked for by $AWFM_GPU_EXACT_LOOKUP=1; advisor, round 5) */
    if (!lengthTable && forced && !(mixedEnv && atoi(mixedEnv) == 0)) lengthTable = ensureLengthTables(g);
    if (!lengthTable && !off) return 0;
  }
  std::lock_guard<std::mutex> lock(g->orderMutex);
  /* [counters 256 B: the leftover count][leftover list: nq x 8 B] */
  const size_t listAt = 256u, total = listAt + alignUp256(nq * 8u);
#define EXACT_TRY(call)                     \
  do {                                      \
    hipError_t e__ = (call);                \
    if (e__ != hipSuccess) {                \
      setError(#call, e__);                 \
      return -(int)AwFmGeneralFailure;      \
    }                                       \
  } while (0)
  EXACT_TRY(orderBeginSlot(g, s));
  OrderSlotScope slotScope(g, s);
  if (!ensureOrderScratch(g, total)) return kOrderNoScratch;
  uint8_t *w = (uint8_t *)g->dOrder;
  unsigned *leftoverCount = (unsigned *)w;
  unsigned long long *leftover = (unsigned long long *)(w + listAt);
  EXACT_TRY(hipMemsetAsync(w, 0, 256, s));
  const bool pairOff = !pairSteps(g);
  EXACT_TRY(awfmGpuLaunchExactLookup(g, s, nullptr, nullptr, lengthTable, dChars, off, fixedLength, nq, pairOff, rng, dCounts, leftover, leftoverCount));
  /* what the lookup kernel left, letter by letter (exact by construction): the last kernel of the search carries the event
   * that says the scratch slot is free again */
  {
    const bool narrow = awfmImageNarrow(g);
    const unsigned long long *csr = (const unsigned long long *)off;
    enum AwFmReturnCode left;
    if (off) left = narrow ? launchLeftover<true, true>(g, s, dChars, csr, fixedLength, nq, rng, dCounts, leftover, 8u, 0u, leftoverCount, nullptr)
                           : launchLeftover<false, true>(g, s, dChars, csr, fixedLength, nq, rng, dCounts, leftover, 8u, 0u, leftoverCount, nullptr);
    else left = narrow ? launchLeftover<true, false>(g, s, dChars, nullptr, fixedLength, nq, rng, dCounts, leftover, 8u, 0u, leftoverCount, nullptr)
                       : launchLeftover<false, false>(g, s, dChars, nullptr, fixedLength, nq, rng, dCounts, leftover, 8u, 0u, leftoverCount, nullptr);
    if (left != AwFmSuccess) return -(int)left;
  }
  EXACT_TRY(slotScope.end());
#undef EXACT_TRY
  return 1;
}

/* ------------------------------------------------------------------ compulsory-traffic tally of the seed-order search */

namespace {
__global__ void __launch_bounds__(256) popcountWordsKernel(const unsigned long long *__restrict__ words, unsigned long long n,
                                                           unsigned long long *__restrict__ total) {
  unsigned long long sum = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256ull)
    sum += (unsigned long long)__popcll(words[i]);
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
  if ((threadIdx.x & 63u) == 0u && sum) atomicAdd(total, sum);
}
}  // namespace

/* see include/awfm_gpu.h */
extern "C" enum AwFmReturnCode awfmGpuMixedLookupLineTally(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                                           uint64_t numQueries, uint64_t tallyOut[8]) {
  if (!g || !dChars || !dOffsets || !tallyOut) {
    setError("awfmGpuMixedLookupLineTally: null argument");
    return AwFmNullPtrError;
  }
  for (int i = 0; i < 8; i++) tallyOut[i] = 0;
  const bool capable = !g->amino && g->dev.deepSeed && g->dev.deepNarrow == (awfmImageNarrow(g) ? 1u : 2u) && g->dev.deepK >= 2u &&
                       g->dev.deepK <= 16u && g->dev.seedK < g->dev.deepK;
  DeviceGuard guard(g->device);
  const uint2 *lengthTable = capable ? ensureLengthTables(g) : nullptr;
  if (!lengthTable) {
    setError("awfmGpuMixedLookupLineTally: this image has no tables per k-mer length (a nucleotide image with its deeper table in 8-byte entries)");
    return AwFmUnsupportedVersionError;
  }
  const uint64_t lengthWords = (awfmLengthTableAt(g->dev.deepK) * 8u / 128u + 64u) / 64u;
  const uint64_t deepWords = ((1ull << (2u * g->dev.deepK)) * 8u / 128u + 64u) / 64u;
  const uint64_t pairWords = (g->numBlocks + 64u) / 64u, nucWords = (g->numBlocks / 2u + 64u) / 64u;
  const unsigned levels = awfmGpuMixedTouchLevels();
  const uint64_t words = lengthWords + deepWords + (uint64_t)levels * (pairWords + nucWords) + 8u;
  unsigned long long *bits = nullptr;
  if (hipMalloc((void **)&bits, words * 8u) != hipSuccess) {
    (void)hipGetLastError();
    setError("awfmGpuMixedLookupLineTally: no device memory for the line bitmaps");
    return AwFmAllocationFailure;
  }
  hipStream_t s = nullptr;
  unsigned long long *pairLines = bits + lengthWords + deepWords, *nucLines = pairLines + (uint64_t)levels * pairWords;
  unsigned long long *sums = nucLines + (uint64_t)levels * nucWords; /* [0..2] the kernel's counts, [3..6] the four line totals */
  const bool pairOff = !pairSteps(g);
  const unsigned useNext = (g->dev.deepNext != 0u && !pairOff ? 1u : 0u) | (pairOff ? 2u : 0u);
  hipError_t e = hipMemsetAsync(bits, 0, words * 8u, s);
  unsigned long long host[8] = {0};
  if (e == hipSuccess)
    e = awfmGpuLaunchMixedTally(g, s, lengthTable, dChars, (const unsigned long long *)dOffsets, (unsigned long long)numQueries, useNext, bits,
                                lengthWords, deepWords, pairWords, nucWords);
  if (e == hipSuccess) {
    const struct { const unsigned long long *from; uint64_t n; } parts[4] = {
        {bits, lengthWords}, {bits + lengthWords, deepWords}, {pairLines, (uint64_t)levels * pairWords}, {nucLines, (uint64_t)levels * nucWords}};
    for (int i = 0; i < 4 && e == hipSuccess; i++) {
      hipLaunchKernelGGL(popcountWordsKernel, dim3(2048), dim3(256), 0, s, parts[i].from, (unsigned long long)parts[i].n, sums + 3 + i);
      e = hipGetLastError();
    }
  }
  if (e == hipSuccess) e = hipMemcpyAsync(host, sums, sizeof host, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(bits);
  if (e != hipSuccess) {
    setError("awfmGpuMixedLookupLineTally", e);
    return AwFmGeneralFailure;
  }
  tallyOut[0] = host[3];
  tallyOut[1] = host[4];
  tallyOut[2] = host[5];
  tallyOut[3] = host[6];
  tallyOut[4] = host[0];
  tallyOut[5] = host[1];
  tallyOut[6] = host[2];
  tallyOut[7] = host[7];
  return AwFmSuccess;
}

/* see include/awfm_gpu.h */
extern "C" enum AwFmReturnCode awfmGpuSearchHitsLineTally(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                                          uint32_t fixedLength, uint64_t numQueries, uint64_t tallyOut[8]) {
  if (!g || !dChars || !tallyOut) {
    setError("awfmGpuSearchHitsLineTally: null argument");
    return AwFmNullPtrError;
  }
  for (int i = 0; i < 8; i++) tallyOut[i] = 0;
  if (!awfmGpuSearchHitsIsOrdered(g, dOffsets != nullptr, fixedLength, numQueries)) {
    setError("awfmGpuSearchHitsLineTally: this batch does not take the seed-order path on this image");
    return AwFmUnsupportedVersionError;
  }
  DeviceGuard guard(g->device);
  const uint64_t seedWords = (g->dev.seedLen * 16u / 128u + 64u) / 64u;
  const uint64_t deepWords = g->dev.deepK ? ((1ull << (2u * g->dev.deepK)) * (g->dev.deepNarrow ? 8u : 16u) / 128u + 64u) / 64u : 1u;
  const uint64_t pairWords = (g->numBlocks + 64u) / 64u, nucWords = (g->numBlocks / 2u + 64u) / 64u;
  const uint64_t words = seedWords + deepWords + (uint64_t)kTouchLevels * (pairWords + nucWords) + 8u;
  unsigned long long *bits = nullptr;
  if (hipMalloc((void **)&bits, words * 8u) != hipSuccess) {
    (void)hipGetLastError();
    setError("awfmGpuSearchHitsLineTally: no device memory for the line bitmaps");
    return AwFmAllocationFailure;
  }
  hipStream_t s = nullptr;
  enum AwFmReturnCode rc = AwFmSuccess;
  OrderTouch touch;
  touch.seedLines = bits;
  touch.deepLines = touch.seedLines + seedWords;
  touch.pairLines = touch.deepLines + deepWords;
  touch.nucLines = touch.pairLines + (uint64_t)kTouchLevels * pairWords;
  touch.pairWords = pairWords;
  touch.nucWords = nucWords;
  unsigned long long *sums = touch.nucLines + (uint64_t)kTouchLevels * nucWords; /* 8 words: hits, then the four line totals */
  touch.hits = sums;
  uint64_t recordBytes = 0;
  hipError_t e = hipMemsetAsync(bits, 0, words * 8u, s);
  if (e == hipSuccess) {
    const int did = orderedSearch(g, s, dChars, (const unsigned long long *)dOffsets, fixedLength, numQueries, nullptr, nullptr, false,
                                  false, &touch, &recordBytes, nullptr);
    if (did <= 0) rc = did < 0 ? (enum AwFmReturnCode)(-did) : AwFmGeneralFailure;
  }
  unsigned generalCount = 0;
  unsigned long long host[8] = {0};
  if (e == hipSuccess && rc == AwFmSuccess) {
    const struct { const unsigned long long *from; uint64_t n; } parts[4] = {
        {touch.seedLines, seedWords}, {touch.deepLines, deepWords}, {touch.pairLines, (uint64_t)kTouchLevels * pairWords},
        {touch.nucLines, (uint64_t)kTouchLevels * nucWords}};
    for (int i = 0; i < 4; i++)
      hipLaunchKernelGGL(popcountWordsKernel, dim3(1024), dim3(256), 0, s, parts[i].from, (unsigned long long)parts[i].n, sums + 1 + i);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host, sums, sizeof(host), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&generalCount, g->dOrder, sizeof(generalCount), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
  }
  (void)hipFree(bits);
  if (e != hipSuccess) {
    setError("awfmGpuSearchHitsLineTally", e);
    return AwFmGeneralFailure;
  }
  if (rc != AwFmSuccess) return rc;
  tallyOut[0] = host[1];                      /* 128-B lines of the index's seed table */
  tallyOut[1] = host[2];                      /* ... of the deeper device-only table */
  tallyOut[2] = host[3];                      /* (level, line) pairs of the pair image */
  tallyOut[3] = host[4];                      /* (level, line) pairs of the one-letter image */
  tallyOut[4] = numQueries - generalCount;    /* k-mers the seed-order kernel searched */
  tallyOut[5] = recordBytes;                  /* bytes of sorted record (+ key) it reads per k-mer */
  tallyOut[6] = host[0];                      /* k-mers with hits (each stores its result) */
  tallyOut[7] = generalCount;                 /* k-mers left to the general kernel (ambiguity characters, > 32 characters) */
  return AwFmSuccess;
}

/* ------------------------------------------------------------------ sparse results */

namespace {
/* the list of the k-mers with hits out of dense results, in batch order: entry flagOffsets[i] when counts[i] != 0 */
__global__ void __launch_bounds__(256) compactDenseKernel(const unsigned *__restrict__ counts, const ulonglong2 *__restrict__ ranges,
                                                          const unsigned long long *__restrict__ flagOffsets, unsigned long long n,
                                                          unsigned cap, unsigned *__restrict__ kmers, ulonglong2 *__restrict__ outRanges,
                                                          unsigned *__restrict__ count) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256ull) {
    if (counts[i] != 0u) {
      const unsigned long long at = flagOffsets[i];
      if (at < cap) {
        kmers[at] = (unsigned)i;
        outRanges[at] = ranges[i];
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *count = (unsigned)(flagOffsets[n] < 0xFFFFFFFFull ? flagOffsets[n] : 0xFFFFFFFFull);
}
}  // namespace

/* see include/awfm_gpu.h */
extern "C" enum AwFmReturnCode awfmGpuSearchHitsCompact(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                                        uint32_t fixedLength, uint64_t numQueries, int packed, uint32_t *dHitKmers,
                                                        struct AwFmSearchRange *dHitRanges, uint32_t capacity, uint32_t *dNumHits,
                                                        void *stream) {
  if (!g || !dChars || !dHitKmers || !dHitRanges || !dNumHits) {
    setError("awfmGpuSearchHitsCompact: null argument");
    return AwFmNullPtrError;
  }
  if (capacity == 0) {
    setError("awfmGpuSearchHitsCompact: the list needs a capacity");
    return AwFmIllegalPositionError;
  }
  if ((g->amino && (packed || dOffsets)) || (!g->amino && !(g->kernel == AWFM_GPU_KERNEL_AUTO || g->kernel == AWFM_GPU_KERNEL_GROUP4)) || (packed && dOffsets)) {
    setError("awfmGpuSearchHitsCompact: this batch does not take the seed-order path on this image");
    return AwFmUnsupportedVersionError;
  }
  DeviceGuard guard(g->device);
  hipStream_t s = (hipStream_t)stream;
  SparseOut sparse; /* (the search path the batch takes fills the list first: fillSparseList / lookupPrepKernel) */
  sparse.count = (unsigned *)dNumHits;
  sparse.cap = capacity;
  sparse.kmers = (unsigned *)dHitKmers;
  sparse.ranges = (ulonglong2 *)dHitRanges;
  const int did = g->amino ? aminoLookupSearch(g, s, dChars, fixedLength, numQueries, nullptr, nullptr, false, &sparse)
                           : orderedSearch(g, s, dChars, (const unsigned long long *)dOffsets, fixedLength, numQueries, nullptr, nullptr,
                                           packed != 0, false, nullptr, nullptr, &sparse);
  if (did < 0) return (enum AwFmReturnCode)(-did);
  if (did == 0) {
    setError("awfmGpuSearchHitsCompact: this batch does not take the seed-order path on this image");
    return AwFmUnsupportedVersionError;
  }
  return AwFmSuccess;
}

/* see include/awfm_gpu.h */
extern "C" enum AwFmReturnCode awfmGpuSearchHitsInOrder(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                                        uint32_t fixedLength, uint64_t numQueries, int packed, uint32_t *dOrderKmers,
                                                        struct AwFmSearchRange *dOrderRanges, void *stream) {
  return awfmGpuSearchHitsInOrderCounts(g, dChars, dOffsets, fixedLength, numQueries, packed, dOrderKmers, dOrderRanges, nullptr, stream);
}

extern "C" enum AwFmReturnCode awfmGpuSearchHitsInOrderCounts(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
                                                              uint32_t fixedLength, uint64_t numQueries, int packed, uint32_t *dOrderKmers,
                                                              struct AwFmSearchRange *dOrderRanges, uint32_t *dOrderCounts, void *stream) {
  if (!g || !dChars || !dOrderKmers || !dOrderRanges) {
    setError("awfmGpuSearchHitsInOrder: null argument");
    return AwFmNullPtrError;
  }
  if (g->amino || !(g->kernel == AWFM_GPU_KERNEL_AUTO || g->kernel == AWFM_GPU_KERNEL_GROUP4) || (packed && dOffsets)) {
    setError("awfmGpuSearchHitsInOrder: this batch does not take the seed-order path on this image");
    return AwFmUnsupportedVersionError;
  }
  DeviceGuard guard(g->device);
  SparseOut out;
  out.count = nullptr;
  out.cap = 0;
  out.kmers = (unsigned *)dOrderKmers;
  out.ranges = (ulonglong2 *)dOrderRanges;
  const int did = orderedSearch(g, (hipStream_t)stream, dChars, (const unsigned long long *)dOffsets, fixedLength, numQueries, nullptr,
                                dOrderCounts, packed != 0, false, nullptr, nullptr, &out);
  if (did < 0) return (enum AwFmReturnCode)(-did);
  if (did == 0) {
    setError("awfmGpuSearchHitsInOrder: this batch does not take the seed-order path on this image");
    return AwFmUnsupportedVersionError;
  }
  return AwFmSuccess;
}

/* ------------------------------------------------------------------ seed-bucket sharding (round 6; include/awfm_gpu.h) */

/* the table a fixed-length batch of `totalQueries` k-mers starts from on this image, and the format of its 8-byte records;
 * false: such a batch is not one for bucketed records */
static bool shardedFormat(const AwFmGpuIndex *g, uint32_t fixedLength, uint64_t totalQueries, unsigned *depthOut, const ulonglong2 **tableOut,
                          BucketFormat *fmtOut) {
  if (g->amino || !(g->kernel == AWFM_GPU_KERNEL_AUTO || g->kernel == AWFM_GPU_KERNEL_GROUP4)) return false;
  if (totalQueries == 0 || totalQueries >= 0xFFFFFFFFull || g->dev.seedK == 0 || g->dev.seedK >= 32 || g->dev.deepK >= 32) return false;
  if (fixedLength == 0 || fixedLength > 32) return false;
  const bool deep = g->dev.deepK != 0 && fixedLength >= g->dev.deepK;
  const unsigned depth = deep ? g->dev.deepK : g->dev.seedK;
  if (fixedLength < depth) return false;
  const BucketFormat fmt = bucketFormat(depth, totalQueries);
  if (!bucketFits(fixedLength, fmt)) return false;
  *depthOut = depth;
  *tableOut = deep ? g->dev.deepSeed : g->dev.seed;
  *fmtOut = fmt;
  return true;
}

extern "C" uint32_t awfmGpuOrderBuckets(const AwFmGpuIndex *g, uint32_t fixedLength, uint64_t totalQueries) {
  unsigned depth = 0;
  const ulonglong2 *table = nullptr;
  BucketFormat fmt;
  if (!g || !shardedFormat(g, fixedLength, totalQueries, &depth, &table, &fmt)) return 0;
  return 1u << fmt.bucketBits;
}

extern "C" enum AwFmReturnCode awfmGpuOrderKmers(AwFmGpuIndex *g, const uint8_t *dChars, uint32_t fixedLength, uint64_t numQueries,
                                                 uint64_t firstNumber, uint64_t totalQueries, uint64_t *dRecords, uint32_t *dBucketStart,
                                                 void *stream) {
  if (!g || !dChars || !dRecords || !dBucketStart) {
    setError("awfmGpuOrderKmers: null argument");
    return AwFmNullPtrError;
  }
  unsigned depth = 0;
  const ulonglong2 *table = nullptr;
  BucketFormat fmt;
  if (numQueries == 0 || firstNumber + numQueries > totalQueries || !shardedFormat(g, fixedLength, totalQueries, &depth, &table, &fmt)) {
    setError("awfmGpuOrderKmers: a shard [first, first + n) of a fixed-length nucleotide batch whose 8-byte records fit (awfmGpuOrderBuckets)");
    return AwFmIllegalPositionError;
  }
  DeviceGuard guard(g->device);
  hipStream_t s = (hipStream_t)stream;
  std::lock_guard<std::mutex> lock(g->orderMutex);
  const unsigned long long nq = numQueries;
  const unsigned bins = (1u << fmt.bucketBits) + 1u, binsPad = (bins + 3u) & ~3u;
  /* [counters 128 KB][hist -> sub-run starts: 8 shares x binsPad][cursors: 8 x binsPad][codes: nq x 8] of the image's scratch */
  const size_t histAt = kOrderCounterBytes, cursorsAt = histAt + alignUp256((size_t)kShares * binsPad * 4u);
  const size_t codesAt = cursorsAt + alignUp256((size_t)kShares * binsPad * 4u), total = codesAt + alignUp256(nq * 8u);
#define SHARD_TRY(call)                     \
  do {                                      \
    hipError_t e__ = (call);                \
    if (e__ != hipSuccess) {                \
      setError(#call, e__);                 \
      return AwFmGeneralFailure;            \
    }                                       \
  } while (0)
  SHARD_TRY(orderBeginSlot(g, s));
  OrderSlotScope slotScope(g, s);
  if (!ensureOrderScratch(g, total)) return AwFmAllocationFailure;
  uint8_t *w = (uint8_t *)g->dOrder;
  unsigned *generalCount = (unsigned *)w, *hist = (unsigned *)(w + histAt), *cursors = (unsigned *)(w + cursorsAt);
  unsigned long long *codes = (unsigned long long *)(w + codesAt);
  SHARD_TRY(hipMemsetAsync(w, 0, codesAt, s));
  const unsigned long long perShare256 = (shareSize(nq) + 255ull) / 256ull;
  unsigned encodeGrid = (unsigned)(perShare256 * kShares < (unsigned long long)g->numCUs * 8u ? perShare256 * kShares : (unsigned long long)g->numCUs * 8u);
  encodeGrid = (encodeGrid + kShares - 1u) / kShares * kShares;
  launchEncode4(fixedLength, encodeGrid, bins * 4u, s, dChars, fmt, nq, codes, hist, binsPad);
  SHARD_TRY(hipGetLastError());
  hipLaunchKernelGGL(bucketScanSharesKernel, dim3(1), dim3(1024), 0, s, hist, bins, binsPad, (unsigned *)dBucketStart, generalCount, (unsigned)nq);
  SHARD_TRY(hipGetLastError());
  const size_t partitionLds = (size_t)kPartitionTile * 8u + 3u * binsPad * 4u;
  SHARD_TRY(hipFuncSetAttribute((const void *)partitionKernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(kPartitionTile * 8u + 3u * (((1u << kBucketBitsMax) + 4u) & ~3u) * 4u)));
  const unsigned long long tilesPerShare = shareSize(nq) / kPartitionTile;
  unsigned partitionGrid = (unsigned)(tilesPerShare * kShares < (unsigned long long)g->numCUs ? tilesPerShare * kShares : (unsigned long long)g->numCUs);
  partitionGrid = (partitionGrid + kShares - 1u) / kShares * kShares;
  /* (the records carry the k-mers' numbers in the WHOLE batch: firstNumber + the number in this shard) */
  hipLaunchKernelGGL(partitionKernel, dim3(partitionGrid), dim3(kPartitionThreads), partitionLds, s, (const unsigned long long *)codes, fixedLength, fmt, nq,
                     (const unsigned *)hist, cursors, (unsigned long long *)dRecords, 1u, (const unsigned *)nullptr, (const unsigned *)nullptr,
                     (const unsigned *)nullptr, 0u, (unsigned long long)firstNumber);
  SHARD_TRY(hipGetLastError());
  /* (the size of the last bin -- the k-mers awfmGpuSearchGeneralRecords takes -- behind the bucket starts) */
  SHARD_TRY(hipMemcpyAsync(dBucketStart + bins + 1u, generalCount, 4, hipMemcpyDeviceToDevice, s));
  SHARD_TRY(slotScope.end());
  return AwFmSuccess;
}

/* ---- what a rank holds after the exchange, put in bucket order (include/awfm_gpu.h) ---- */
namespace {
constexpr unsigned kMergeParts = 16; /* workgroups per bucket */
/* Workgroup (b, part): the part-th share of every slice's run of bucket b, copied to where the bucket's runs lie next to each
 * other, slice by slice.  Where that is needs no pass of its own: the records before bucket b are the sum over the slices of
 * THEIR records before b (starts[j][b], a slice's starts being relative to the slice).  The workgroups behind the last bucket
 * write the bucket starts awfmGpuSearchOrderedRecords wants. */
__global__ void __launch_bounds__(256)
    mergeBucketRunsKernel(const unsigned long long *__restrict__ received, const unsigned long long *__restrict__ sliceAt,
                          const unsigned *__restrict__ starts, const unsigned numSlices, const unsigned firstBucket, const unsigned endBucket,
                          const unsigned buckets, unsigned long long *__restrict__ out, unsigned *__restrict__ bucketStartOut) {
  const unsigned nb = endBucket - firstBucket, stride = nb + 1u;
  const unsigned b = blockIdx.x / kMergeParts, part = blockIdx.x % kMergeParts;
  if (b >= nb) { /* the last workgroups: the array of bucket starts (buckets + 3 words) */
    const unsigned worker = (blockIdx.x - nb * kMergeParts) * 256u + threadIdx.x, workers = (gridDim.x - nb * kMergeParts) * 256u;
    unsigned long long total = 0;
    for (unsigned j = 0; j < numSlices; j++) total += starts[j * stride + nb];
    for (unsigned e = worker; e < buckets + 3u; e += workers) {
      unsigned long long v = 0;
      if (e >= firstBucket && e <= endBucket) {
        for (unsigned j = 0; j < numSlices; j++) v += starts[j * stride + (e - firstBucket)];
      } else if (e > endBucket && e < buckets + 2u) {
        v = total; /* nothing behind this rank's buckets; [buckets], [buckets + 1]: the records held */
      }
      bucketStartOut[e] = (unsigned)v;
    }
    return;
  }
  unsigned long long to = 0;
  for (unsigned j = 0; j < numSlices; j++) to += starts[j * stride + b];
  for (unsigned j = 0; j < numSlices; j++) {
    const unsigned first = starts[j * stride + b], count = starts[j * stride + b + 1u] - first;
    const unsigned share = (count + kMergeParts - 1u) / kMergeParts;
    const unsigned lo = part * share < count ? part * share : count, hi = lo + share < count ? lo + share : count;
    const unsigned long long *from = received + sliceAt[j] + first;
    for (unsigned i = lo + threadIdx.x; i < hi; i += 256u) out[to + i] = from[i];
    to += count;
  }
}
}  // namespace

extern "C" enum AwFmReturnCode awfmGpuMergeBucketRuns(AwFmGpuIndex *g, const uint64_t *dReceived, const uint64_t *dSliceAt, const uint32_t *dSliceStarts,
                                                      uint32_t numSlices, uint32_t firstBucket, uint32_t endBucket, uint32_t buckets, uint64_t *dRecords,
                                                      uint32_t *dBucketStart, void *stream) {
  if (!g || !dReceived || !dSliceAt || !dSliceStarts || !dRecords || !dBucketStart) {
    setError("awfmGpuMergeBucketRuns: null argument");
    return AwFmNullPtrError;
  }
  if (numSlices == 0u || firstBucket >= endBucket || endBucket > buckets) {
    setError("awfmGpuMergeBucketRuns: slices of the buckets [first, end) of `buckets` (awfmGpuOrderBuckets)");
    return AwFmIllegalPositionError;
  }
  DeviceGuard guard(g->device);
  const unsigned nb = endBucket - firstBucket;
  hipLaunchKernelGGL(mergeBucketRunsKernel, dim3(nb * kMergeParts + 4u), dim3(256), 0, (hipStream_t)stream, (const unsigned long long *)dReceived,
                     (const unsigned long long *)dSliceAt, dSliceStarts, numSlices, firstBucket, endBucket, buckets, (unsigned long long *)dRecords, dBucketStart);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    setError("awfmGpuMergeBucketRuns", e);
    return AwFmGeneralFailure;
  }
  return AwFmSuccess;
}

extern "C" enum AwFmReturnCode awfmGpuSearchOrderedRecords(AwFmGpuIndex *g, const uint64_t *dRecords, const uint32_t *dBucketStart, uint32_t firstBucket,
                                                           uint32_t endBucket, uint32_t fixedLength, uint64_t totalQueries, uint32_t *dOrderKmers,
                                                           struct AwFmSearchRange *dOrderRanges, void *stream) {
  return awfmGpuSearchOrderedRecordsCounts(g, dRecords, dBucketStart, firstBucket, endBucket, fixedLength, totalQueries, dOrderKmers, dOrderRanges, nullptr,
                                           stream);
}

extern "C" enum AwFmReturnCode awfmGpuSearchOrderedRecordsCounts(AwFmGpuIndex *g, const uint64_t *dRecords, const uint32_t *dBucketStart,
                                                                 uint32_t firstBucket, uint32_t endBucket, uint32_t fixedLength, uint64_t totalQueries,
                                                                 uint32_t *dOrderKmers, struct AwFmSearchRange *dOrderRanges, uint32_t *dOrderCounts,
                                                                 void *stream) {
  if (!g || !dRecords || !dBucketStart || !dOrderKmers || !dOrderRanges) {
    setError("awfmGpuSearchOrderedRecords: null argument");
    return AwFmNullPtrError;
  }
  unsigned depth = 0;
  const ulonglong2 *table = nullptr;
  BucketFormat fmt;
  if (!shardedFormat(g, fixedLength, totalQueries, &depth, &table, &fmt) || firstBucket >= endBucket || endBucket > (1u << fmt.bucketBits)) {
    setError("awfmGpuSearchOrderedRecords: buckets [first, end) of the records awfmGpuOrderKmers makes for this image and batch");
    return AwFmIllegalPositionError;
  }
  DeviceGuard guard(g->device);
  hipStream_t s = (hipStream_t)stream;
  std::lock_guard<std::mutex> lock(g->orderMutex);
  fmt.firstBucket = firstBucket;
  fmt.endBucket = endBucket;
  /* the ticket counters of the search kernel: the counter block of the image's scratch */
  if (orderBeginSlot(g, s) != hipSuccess) {
    setError("seed-order search: could not order the use of its scratch across streams");
    return AwFmGeneralFailure;
  }
  OrderSlotScope slotScope(g, s);
  if (!ensureOrderScratch(g, kOrderCounterBytes)) return AwFmAllocationFailure;
  unsigned *generalCount = (unsigned *)g->dOrder;
  AWFM_HIP_TRY(hipMemsetAsync(generalCount, 0, kOrderCounterBytes, s), AwFmGeneralFailure);
  for (int i = 0; i < 4; i++) g->orderTiming[i] = nullptr;
  SparseOut out;
  out.count = nullptr;
  out.cap = 0;
  out.kmers = (unsigned *)dOrderKmers;
  out.ranges = (ulonglong2 *)dOrderRanges;
  const bool pair = pairSteps(g);
  enum AwFmReturnCode rc;
#define SHARD_GO(NR, PR) \
  launchOrderedKernel<NR, false, PR, false, true>(g, s, fixedLength, depth, table, totalQueries, dRecords, generalCount, nullptr, dOrderCounts, nullptr, (const unsigned *)dBucketStart, fmt, &out)
  if (awfmImageNarrow(g)) rc = pair ? SHARD_GO(true, true) : SHARD_GO(true, false);
  else rc = pair ? SHARD_GO(false, true) : SHARD_GO(false, false);
#undef SHARD_GO
  if (rc != AwFmSuccess) return rc;
  AWFM_HIP_TRY(slotScope.end(), AwFmGeneralFailure);
  return AwFmSuccess;
}

extern "C" enum AwFmReturnCode awfmGpuSearchGeneralRecords(AwFmGpuIndex *g, const uint8_t *dChars, uint32_t fixedLength, uint64_t numQueries,
                                                           uint64_t firstNumber, uint64_t totalQueries, const uint64_t *dRecords,
                                                           const uint32_t *dBucketStart, uint32_t *dOrderKmers, struct AwFmSearchRange *dOrderRanges,
                                                           void *stream) {
  if (!g || !dChars || !dRecords || !dBucketStart || !dOrderKmers || !dOrderRanges) {
    setError("awfmGpuSearchGeneralRecords: null argument");
    return AwFmNullPtrError;
  }
  unsigned depth = 0;
  const ulonglong2 *table = nullptr;
  BucketFormat fmt;
  if (numQueries == 0 || firstNumber + numQueries > totalQueries || !shardedFormat(g, fixedLength, totalQueries, &depth, &table, &fmt)) {
    setError("awfmGpuSearchGeneralRecords: the shard awfmGpuOrderKmers was given");
    return AwFmIllegalPositionError;
  }
  DeviceGuard guard(g->device);
  hipStream_t s = (hipStream_t)stream;
  /* the last bin of the shard's records: the k-mers with ambiguity characters, a record each = its number in the whole batch.
   * The general kernel reads the k-mer under that number: the character array as if it began at the batch's first k-mer.
   * Entries [dBucketStart[buckets], numQueries) of the order arrays get {number, range}. */
  const unsigned bins = (1u << fmt.bucketBits) + 1u;
  const unsigned *leftCount = (const unsigned *)dBucketStart + bins + 1u; /* bucketScanSharesKernel leaves the last bin's size there */
  SparseOut out;
  out.count = nullptr;
  out.cap = 0;
  out.kmers = (unsigned *)dOrderKmers;
  out.ranges = (ulonglong2 *)dOrderRanges;
  const uint8_t *base = dChars - (size_t)firstNumber * fixedLength;
  const bool narrow = awfmImageNarrow(g);
  const unsigned grid = narrow ? residentGrid(g, searchKernel<false, 4, false, false, true, true>) : residentGrid(g, searchKernel<false, 4, false, false, false, true>);
  if (narrow)
    hipLaunchKernelGGL((searchKernel<false, 4, false, false, true, true>), dim3(grid), dim3(kThreads), 0, s, g->dev, base, (const unsigned long long *)nullptr, fixedLength,
                       (unsigned long long)(firstNumber + numQueries), (ulonglong2 *)nullptr, (unsigned *)nullptr, (unsigned long long *)nullptr,
                       (const unsigned char *)dRecords, 8u, 0u, (unsigned long long)numQueries, leftCount, out, (const unsigned *)nullptr, 0u);
  else
    hipLaunchKernelGGL((searchKernel<false, 4, false, false, false, true>), dim3(grid), dim3(kThreads), 0, s, g->dev, base, (const unsigned long long *)nullptr, fixedLength,
                       (unsigned long long)(firstNumber + numQueries), (ulonglong2 *)nullptr, (unsigned *)nullptr, (unsigned long long *)nullptr,
                       (const unsigned char *)dRecords, 8u, 0u, (unsigned long long)numQueries, leftCount, out, (const unsigned *)nullptr, 0u);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  return AwFmSuccess;
}
#undef SHARD_TRY

extern "C" enum AwFmReturnCode awfmGpuCompactHits(AwFmGpuIndex *g, const uint32_t *dCounts, const struct AwFmSearchRange *dRanges,
                                                  uint64_t numQueries, uint64_t *dFlagOffsets, void *dScratch, uint32_t *dHitKmers,
                                                  struct AwFmSearchRange *dHitRanges, uint32_t capacity, uint32_t *dNumHits,
                                                  void *stream) {
  if (!g || !dCounts || !dRanges || !dFlagOffsets || !dScratch || !dHitKmers || !dHitRanges || !dNumHits) {
    setError("awfmGpuCompactHits: null argument");
    return AwFmNullPtrError;
  }
  if (numQueries == 0 || numQueries >= 0xFFFFFFFFull) {
    setError("awfmGpuCompactHits: 1 .. 2^32 - 2 k-mers");
    return AwFmIllegalPositionError;
  }
  DeviceGuard guard(g->device);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(fillSparseKernel, dim3((capacity + 255u) / 256u), dim3(256), 0, s, (unsigned *)dHitKmers, (ulonglong2 *)dHitRanges,
                     (unsigned)capacity, (unsigned *)dNumHits);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  const enum AwFmReturnCode rc = awfmGpuScanFlags(g, dCounts, numQueries, dFlagOffsets, dScratch, s);
  if (rc != AwFmSuccess) return rc;
  hipLaunchKernelGGL(compactDenseKernel, dim3((unsigned)g->numCUs * 8u), dim3(256), 0, s, (const unsigned *)dCounts,
                     (const ulonglong2 *)dRanges, (const unsigned long long *)dFlagOffsets, (unsigned long long)numQueries,
                     (unsigned)capacity, (unsigned *)dHitKmers, (ulonglong2 *)dHitRanges, (unsigned *)dNumHits);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  return AwFmSuccess;
}

extern "C" enum AwFmReturnCode awfmGpuSortHits(AwFmGpuIndex *g, uint32_t *dHitKmers, struct AwFmSearchRange *dHitRanges,
                                               uint32_t numEntries, void *stream) {
  if (!g || !dHitKmers || !dHitRanges) {
    setError("awfmGpuSortHits: null argument");
    return AwFmNullPtrError;
  }
  if (numEntries < 2) return AwFmSuccess;
  DeviceGuard guard(g->device);
  hipStream_t s = (hipStream_t)stream;
  std::lock_guard<std::mutex> lock(g->orderMutex);
  size_t tempBytes = 0;
  unsigned *nullKeys = nullptr;
  ulonglong2 *nullValues = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, tempBytes, nullKeys, nullKeys, nullValues, nullValues, (size_t)numEntries, 0u, 32u, s) != hipSuccess) {
    setError("awfmGpuSortHits: radix sort sizing failed");
    return AwFmGeneralFailure;
  }
  const size_t keysAt = alignUp256(tempBytes), valuesAt = keysAt + alignUp256((size_t)numEntries * 4u);
  const size_t total = valuesAt + alignUp256((size_t)numEntries * 16u);
  if (total > g->sparseBytes) {
    if (g->dSparse) (void)hipFree(g->dSparse);
    g->dSparse = nullptr;
    g->sparseBytes = 0;
    if (hipMalloc(&g->dSparse, total + total / 4) != hipSuccess) {
      (void)hipGetLastError();
      setError("awfmGpuSortHits: no device memory for the sort");
      return AwFmAllocationFailure;
    }
    g->sparseBytes = total + total / 4;
  }
  /* the temporaries are shared by every sort on this image: order their use across streams (as the searches' scratch is) */
  AWFM_HIP_TRY(gateEnter(g->sparseGate, s), AwFmGeneralFailure);
  uint8_t *w = (uint8_t *)g->dSparse;
  unsigned *keysOut = (unsigned *)(w + keysAt);
  ulonglong2 *valuesOut = (ulonglong2 *)(w + valuesAt);
  AWFM_HIP_TRY(rocprim::radix_sort_pairs(w, tempBytes, (unsigned *)dHitKmers, keysOut, (ulonglong2 *)dHitRanges, valuesOut,
                                         (size_t)numEntries, 0u, 32u, s),
               AwFmGeneralFailure);
  AWFM_HIP_TRY(hipMemcpyAsync(dHitKmers, keysOut, (size_t)numEntries * 4u, hipMemcpyDeviceToDevice, s), AwFmGeneralFailure);
  AWFM_HIP_TRY(hipMemcpyAsync(dHitRanges, valuesOut, (size_t)numEntries * 16u, hipMemcpyDeviceToDevice, s), AwFmGeneralFailure);
  AWFM_HIP_TRY(gateLeave(g->sparseGate, s, false), AwFmGeneralFailure);
  return AwFmSuccess;
}

/* ---- the list of hits put in k-mer order without the host knowing how long it is ----
 * The k-mer numbers of a list are distinct and below numQueries, so the place of an entry in k-mer order is the number of
 * listed k-mers before it: a bitmap of the batch (one bit per k-mer, 12.5 MB for 10^8), a count per 4096 bits, one scan of
 * those counts, and every entry finds its rank by popcounts.  Every kernel has a fixed grid and reads the list's length
 * from the device word the search left it in, so nothing waits for the host (awfmGpuSortHits, a radix sort, needs the
 * length on the host -- a stream synchronisation per batch -- or sorts the list's whole capacity). */
namespace {
constexpr unsigned kRankBlockWords = 64; /* 64-bit words per counted block: 4096 k-mers */
__global__ void __launch_bounds__(256) rankMarkKernel(const unsigned *__restrict__ kmers, const ulonglong2 *__restrict__ ranges,
                                                      const unsigned *__restrict__ count, const unsigned cap, const unsigned long long numQueries,
                                                      unsigned long long *__restrict__ bitmap, unsigned *__restrict__ tmpKmers,
                                                      ulonglong2 *__restrict__ tmpRanges) {
  const unsigned n = *count < cap ? *count : cap;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const unsigned key = kmers[i];
    tmpKmers[i] = key;
    tmpRanges[i] = ranges[i];
    if ((unsigned long long)key < numQueries) atomicOr(bitmap + (key >> 6), 1ull << (key & 63u));
  }
}
/* one wave per block of the bitmap */
__global__ void __launch_bounds__(256) rankBlockCountKernel(const unsigned long long *__restrict__ bitmap, const unsigned numBlocks,
                                                            unsigned *__restrict__ blockCount) {
  const unsigned lane = threadIdx.x & 63u, wavesPerGrid = gridDim.x * 4u;
  for (unsigned b = blockIdx.x * 4u + (threadIdx.x >> 6); b < numBlocks; b += wavesPerGrid) {
    unsigned c = (unsigned)__popcll(bitmap[(unsigned long long)b * kRankBlockWords + lane]);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if (lane == 0) blockCount[b] = c;
  }
}
/* exclusive scan of blockCount[0..numBlocks) in place, one workgroup */
__global__ void __launch_bounds__(1024) rankBlockScanKernel(unsigned *__restrict__ blockCount, const unsigned numBlocks) {
  __shared__ unsigned sWave[16];
  __shared__ unsigned sCarry;
  if (threadIdx.x == 0) sCarry = 0u;
  __syncthreads();
  for (unsigned base = 0; base < numBlocks; base += 1024u) {
    const unsigned e = base + threadIdx.x;
    const unsigned v = e < numBlocks ? blockCount[e] : 0u;
    unsigned incl = v;
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned up = __shfl_up(incl, off);
      if ((int)(threadIdx.x & 63u) >= off) incl += up;
    }
    if ((threadIdx.x & 63u) == 63u) sWave[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned before = sCarry;
    for (unsigned w = 0; w < (threadIdx.x >> 6); w++) before += sWave[w];
    if (e < numBlocks) blockCount[e] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023u) sCarry = before + incl;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) rankPlaceKernel(const unsigned *__restrict__ tmpKmers, const ulonglong2 *__restrict__ tmpRanges,
                                                       const unsigned *__restrict__ count, const unsigned cap, const unsigned long long numQueries,
                                                       const unsigned long long *__restrict__ bitmap, const unsigned *__restrict__ blockStart,
                                                       unsigned *__restrict__ kmers, ulonglong2 *__restrict__ ranges) {
  const unsigned n = *count < cap ? *count : cap;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const unsigned key = tmpKmers[i];
    if ((unsigned long long)key >= numQueries) continue; /* not a k-mer of the batch: the entry stays where the search put it */
    const unsigned word = key >> 6, block = word / kRankBlockWords;
    unsigned rank = blockStart[block];
    for (unsigned w = block * kRankBlockWords; w < word; w++) rank += (unsigned)__popcll(bitmap[w]);
    rank += (unsigned)__popcll(bitmap[word] & ((1ull << (key & 63u)) - 1ull));
    kmers[rank] = key;
    ranges[rank] = tmpRanges[i];
  }
}
}  // namespace

/* see include/awfm_gpu.h */
extern "C" enum AwFmReturnCode awfmGpuSortHitsOnDevice(AwFmGpuIndex *g, uint32_t *dHitKmers, struct AwFmSearchRange *dHitRanges,
                                                       uint32_t capacity, const uint32_t *dNumHits, uint64_t numQueries, void *stream) {
  if (!g || !dHitKmers || !dHitRanges || !dNumHits) {
    setError("awfmGpuSortHitsOnDevice: null argument");
    return AwFmNullPtrError;
  }
  if (capacity == 0 || numQueries == 0 || numQueries >= 0xFFFFFFFFull) {
    setError("awfmGpuSortHitsOnDevice: a list needs a capacity and a batch of 1 .. 2^32 - 2 k-mers");
    return AwFmIllegalPositionError;
  }
  DeviceGuard guard(g->device);
  hipStream_t s = (hipStream_t)stream;
  std::lock_guard<std::mutex> lock(g->orderMutex);
  const unsigned numBlocks = (unsigned)((numQueries + 64ull * kRankBlockWords - 1ull) / (64ull * kRankBlockWords));
  const size_t bitmapBytes = (size_t)numBlocks * kRankBlockWords * 8u;
  const size_t countsAt = alignUp256(bitmapBytes), keysAt = countsAt + alignUp256((size_t)numBlocks * 4u);
  const size_t valuesAt = keysAt + alignUp256((size_t)capacity * 4u), total = valuesAt + alignUp256((size_t)capacity * 16u);
  if (total > g->sparseBytes) {
    if (g->dSparse) (void)hipFree(g->dSparse);
    g->dSparse = nullptr;
    g->sparseBytes = 0;
    if (hipMalloc(&g->dSparse, total + total / 4) != hipSuccess) {
      (void)hipGetLastError();
      setError("awfmGpuSortHitsOnDevice: no device memory for the bitmap of the batch");
      return AwFmAllocationFailure;
    }
    g->sparseBytes = total + total / 4;
  }
  /* the temporaries are shared by every sort on this image: order their use across streams (as the searches' scratch is) */
  AWFM_HIP_TRY(gateEnter(g->sparseGate, s), AwFmGeneralFailure);
  uint8_t *w = (uint8_t *)g->dSparse;
  unsigned long long *bitmap = (unsigned long long *)w;
  unsigned *blockCount = (unsigned *)(w + countsAt), *tmpKmers = (unsigned *)(w + keysAt);
  ulonglong2 *tmpRanges = (ulonglong2 *)(w + valuesAt);
  AWFM_HIP_TRY(hipMemsetAsync(bitmap, 0, bitmapBytes, s), AwFmGeneralFailure);
  const unsigned listGrid = (capacity + 255u) / 256u < (unsigned)g->numCUs * 4u ? (capacity + 255u) / 256u : (unsigned)g->numCUs * 4u;
  hipLaunchKernelGGL(rankMarkKernel, dim3(listGrid), dim3(256), 0, s, (const unsigned *)dHitKmers, (const ulonglong2 *)dHitRanges,
                     (const unsigned *)dNumHits, (unsigned)capacity, (unsigned long long)numQueries, bitmap, tmpKmers, tmpRanges);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  const unsigned countGrid = (numBlocks + 3u) / 4u < (unsigned)g->numCUs * 8u ? (numBlocks + 3u) / 4u : (unsigned)g->numCUs * 8u;
  hipLaunchKernelGGL(rankBlockCountKernel, dim3(countGrid), dim3(256), 0, s, (const unsigned long long *)bitmap, numBlocks, blockCount);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  hipLaunchKernelGGL(rankBlockScanKernel, dim3(1), dim3(1024), 0, s, blockCount, numBlocks);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  /* the last kernel carries the gate's event as its stop event */
  const bool eager = !gateLazy(s);
  AWFM_LAUNCH_WITH_EVENTS(rankPlaceKernel, dim3(listGrid), dim3(256), 0, s, nullptr, eager ? g->sparseGate.done : nullptr, (const unsigned *)tmpKmers,
                        (const ulonglong2 *)tmpRanges, (const unsigned *)dNumHits, (unsigned)capacity, (unsigned long long)numQueries,
                        (const unsigned long long *)bitmap, (const unsigned *)blockCount, (unsigned *)dHitKmers, (ulonglong2 *)dHitRanges);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  AWFM_HIP_TRY(gateLeave(g->sparseGate, s, eager), AwFmGeneralFailure);
  return AwFmSuccess;
}
