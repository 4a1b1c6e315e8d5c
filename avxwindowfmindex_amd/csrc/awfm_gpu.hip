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


namespace {
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
   * 11.4-11.7 ms against 12.8-12.9 ms with the bases read from memory) */
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

extern "C" {

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
  const int lanes = awfmGpuLanesPerQuery(g);
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
    launchSearch<true>(g, plain, awfmGpuLanesPerQuery(g), (hipStream_t)0, dChars, (const unsigned long long *)dOffsets,
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
  enum AwFmReturnCode rc = awfmGpuEnsureWork(g, l.total);
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
  if (const char *env = awfmKnob(AWFM_KNOB_HIT_BUDGET_BYTES)) {
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
  enum AwFmReturnCode rc = awfmGpuEnsureWork(g, l.total);
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

