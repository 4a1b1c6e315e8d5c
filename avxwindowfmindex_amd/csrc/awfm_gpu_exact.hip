/*
 * awfm_gpu_exact.hip -- launch of exactLookupSearchKernel (awfm_exact_lookup_kernel.h: awfmGpuSearch's exact ranges through the
 * device-only tables).  A translation unit of its own, as awfm_gpu_mixed.hip is, so that the seed-order kernels keep their
 * register allocation.  The caller (awfm_gpu_ordered.hip: awfmGpuExactLookupSearch) owns the scratch and the stream order.
 */
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "awfm_exact_lookup_kernel.h"

/* awfm_device.h */
hipError_t awfmGpuLaunchExactLookup(const AwFmGpuIndex *g, hipStream_t s, hipEvent_t start, hipEvent_t stop, const void *lengthTable,
                                    const uint8_t *dChars, const unsigned long long *off, unsigned fixedLength, unsigned long long nq,
                                    bool pairOff, ulonglong2 *rng, unsigned *dCounts, unsigned long long *leftover, unsigned *leftoverCount) {
  /* the superblock bases of the pair image are read from memory, as in mixedLookupSearchKernel: the survivors' slots leave room
   * for 5 workgroups per CU, 12-24 KB of bases in LDS would leave 3 */
  DevIndex dev = g->dev;
  dev.lengthBig = (const unsigned long long *)(g->shares ? g->shares : g)->dLengthBig;
  dev.pairSuperInLds = 0u;
  const bool narrow = awfmImageNarrow(g);
  int perCU = 0;
  const hipError_t asked = narrow ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, exactLookupSearchKernel<true>, 256, 0)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, exactLookupSearchKernel<false>, 256, 0);
  if (asked != hipSuccess || perCU < 1) perCU = 4;
  if (perCU > 8) perCU = 8;
  const unsigned long long rounds = (nq + 1023ull) / 1024ull; /* a workgroup takes 1024 k-mers a round */
  unsigned grid = (unsigned)g->numCUs * (unsigned)perCU;
  if (rounds < grid) grid = (unsigned)rounds;
  if (narrow)
    AWFM_LAUNCH_WITH_EVENTS(exactLookupSearchKernel<true>, dim3(grid ? grid : 1u), dim3(256), 0u, s, start, stop, dev, (const uint2 *)lengthTable, dChars,
                            off, fixedLength, nq, pairOff ? 1u : 0u, rng, dCounts, leftover, leftoverCount);
  else
    AWFM_LAUNCH_WITH_EVENTS(exactLookupSearchKernel<false>, dim3(grid ? grid : 1u), dim3(256), 0u, s, start, stop, dev, (const uint2 *)lengthTable, dChars,
                            off, fixedLength, nq, pairOff ? 1u : 0u, rng, dCounts, leftover, leftoverCount);
  return hipGetLastError();
}
