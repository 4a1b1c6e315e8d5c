/*
 * awfm_gpu_mixed.hip -- launches of the kernels of awfm_mixed_lookup_kernel.h ("lookup first" for mixed-length batches).
 *
 * A translation unit of their own: compiled beside the seed-order kernels (awfm_gpu_ordered.hip) they change THOSE kernels'
 * register allocation (one more VGPR in every orderedSearchKernel variant, 4 instead of 2 spilled in the bucketed pair
 * variant -- the code object's LDS layout is shared), and the planted batch is that kernel's.  The caller
 * (wideBucketedSearch, awfmGpuMixedLookupLineTally) owns the scratch, the stream order and the error reporting.
 */
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "awfm_mixed_lookup_kernel.h"

namespace {
/* the image view of a launch: the tables per k-mer length of an image in the wide format come with their long lengths
 * (DevIndex::lengthBig: owned by the primary, set before the table's pointer is published under its mutex) */
DevIndex viewOf(const AwFmGpuIndex *g) {
  DevIndex dev = g->dev;
  dev.lengthBig = (const unsigned long long *)(g->shares ? g->shares : g)->dLengthBig;
  return dev;
}
}  // namespace

/* awfm_device.h */
hipError_t awfmGpuLaunchMixedSample(const AwFmGpuIndex *g, hipStream_t s, const void *lengthTable, const uint8_t *dChars,
                                    const unsigned long long *off, unsigned long long nq, unsigned useNext, unsigned samples,
                                    unsigned long long *aliveOut, unsigned long long *verdictHost, unsigned searchNumber) {
  hipLaunchKernelGGL(mixedSampleAliveKernel, dim3((samples + 255u) / 256u), dim3(256), 0, s, viewOf(g), (const uint2 *)lengthTable, dChars, off, nq,
                     useNext, samples, aliveOut, verdictHost, searchNumber);
  return hipGetLastError();
}

hipError_t awfmGpuLaunchMixedLookup(const AwFmGpuIndex *g, hipStream_t s, hipEvent_t start, hipEvent_t stop, const void *lengthTable,
                                    const uint8_t *dChars, const unsigned long long *off, unsigned long long nq, unsigned useNext,
                                    bool superInLds, const unsigned *sampleAlive, unsigned chooseOf, ulonglong2 *rng, unsigned *dCounts,
                                    unsigned *sparseCount, unsigned sparseCap, unsigned *sparseKmers, ulonglong2 *sparseRanges,
                                    unsigned long long *leftover, unsigned *leftoverCount, unsigned *kept) {
  const bool narrow = awfmImageNarrow(g);
  if (!narrow) superInLds = false; /* (the 32-bit copy of the bases is the narrow kernels') */
  DevIndex dev = viewOf(g);
  dev.pairSuperInLds = superInLds ? 1u : 0u;
  const size_t lds = superInLds ? (size_t)g->dev.numPairSuper * 64u : 0u;
  SparseOut out;
  out.count = sparseCount;
  out.cap = sparseCap;
  out.kmers = sparseKmers;
  out.ranges = sparseRanges;
  /* persistent grid: what is resident; a workgroup takes 1024 k-mers a round */
  int perCU = 0;
  const hipError_t asked = narrow ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, mixedLookupSearchKernel<true>, 256, lds)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, mixedLookupSearchKernel<false>, 256, lds);
  if (asked != hipSuccess || perCU < 1) perCU = 4;
  if (perCU > 8) perCU = 8;
  const unsigned long long rounds = (nq + 1023ull) / 1024ull;
  unsigned grid = (unsigned)g->numCUs * (unsigned)perCU;
  if (rounds < grid) grid = (unsigned)rounds;
  if (narrow)
    AWFM_LAUNCH_WITH_EVENTS(mixedLookupSearchKernel<true>, dim3(grid ? grid : 1u), dim3(256), (unsigned)lds, s, start, stop, dev,
                            (const uint2 *)lengthTable, dChars, off, nq, useNext, sampleAlive, chooseOf, rng, dCounts, out, leftover, leftoverCount,
                            kept);
  else
    AWFM_LAUNCH_WITH_EVENTS(mixedLookupSearchKernel<false>, dim3(grid ? grid : 1u), dim3(256), (unsigned)lds, s, start, stop, dev,
                            (const uint2 *)lengthTable, dChars, off, nq, useNext, sampleAlive, chooseOf, rng, dCounts, out, leftover, leftoverCount,
                            kept);
  return hipGetLastError();
}

/* bits: [lengthWords][deepWords][kMixedTouchLevels x pairWords][kMixedTouchLevels x nucWords][8 sums], zeroed by the caller */
hipError_t awfmGpuLaunchMixedTally(const AwFmGpuIndex *g, hipStream_t s, const void *lengthTable, const uint8_t *dChars,
                                   const unsigned long long *off, unsigned long long nq, unsigned useNext, unsigned long long *bits,
                                   unsigned long long lengthWords, unsigned long long deepWords, unsigned long long pairWords,
                                   unsigned long long nucWords) {
  MixedTouch touch;
  touch.lengthLines = bits;
  touch.deepLines = touch.lengthLines + lengthWords;
  touch.pairLines = touch.deepLines + deepWords;
  touch.nucLines = touch.pairLines + (unsigned long long)kMixedTouchLevels * pairWords;
  touch.pairWords = pairWords;
  touch.nucWords = nucWords;
  touch.sums = touch.nucLines + (unsigned long long)kMixedTouchLevels * nucWords;
  DevIndex dev = viewOf(g);
  dev.pairSuperInLds = 0u;
  if (awfmImageNarrow(g))
    hipLaunchKernelGGL(mixedLookupTallyKernel<true>, dim3((unsigned)g->numCUs * 4u), dim3(256), 0, s, dev, (const uint2 *)lengthTable, dChars, off, nq,
                       useNext, touch);
  else
    hipLaunchKernelGGL(mixedLookupTallyKernel<false>, dim3((unsigned)g->numCUs * 4u), dim3(256), 0, s, dev, (const uint2 *)lengthTable, dChars, off, nq,
                       useNext, touch);
  return hipGetLastError();
}
unsigned awfmGpuMixedTouchLevels(void) { return kMixedTouchLevels; }
