/*
 * awfm_gpu_dense_sa.hip -- the device-only FULL suffix array of an image (a locate becomes one gather per hit instead of a chain of
 * LF steps and a sample read): handed over by the GPU builder, or put together from the sampled array by capped LF walks whose
 * parked ones are completed from each other by pointer jumping; 32-bit entries, or 40-bit ones (DenseSa) for the images that
 * run 64-bit positions.  ref src/AwFmSuffixArray.c:12-18, :179-203, src/AwFmSearch.c:369-427.
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
/* dense device SA construction helpers */
__global__ void iotaKernel(unsigned long long *out, unsigned long long first, unsigned long long count) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = first + i;
}
__global__ void narrowKernel(const unsigned long long *in, unsigned long long count, unsigned *out) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = (unsigned)in[i];
}
}  // namespace

extern "C" {
static enum AwFmReturnCode applyDenseSa(AwFmGpuIndex *g, bool enable, bool capped = false);

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
  awfmGpuAdoptAccelerators(g, true); /* (whatever is still being built behind the first searches) */
  DeviceGuard guard(g->device);
  AwFmGpuLaneLocks lanes(g);
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
struct DenseBuilt {
  void *dDenseSa = nullptr;
  bool denseWide = false;
  uint64_t denseSaBytes = 0;
};
static enum AwFmReturnCode buildDenseSaWide(AwFmGpuIndex *image, bool capped, DenseBuilt *g);
static enum AwFmReturnCode buildDenseSa(AwFmGpuIndex *image, bool capped, DenseBuilt *g);
static enum AwFmReturnCode applyDenseSa(AwFmGpuIndex *g, bool enable, bool capped) {
  (void)awfmGpuSetupSync();
  if (g->dDenseSa) (void)hipFree(g->dDenseSa);
  g->dDenseSa = nullptr;
  g->denseSaBytes = 0;
  g->denseWide = false;
  if (!enable) return AwFmSuccess;
  DenseBuilt built;
  const enum AwFmReturnCode rc = buildDenseSa(g, capped, &built);
  g->dDenseSa = built.dDenseSa;
  g->denseWide = built.denseWide;
  g->denseSaBytes = built.denseSaBytes;
  return rc;
}
/* the construction itself: reads the image (its blocks, its sampled array), writes `g` (what was built) */
static enum AwFmReturnCode buildDenseSa(AwFmGpuIndex *image, bool capped, DenseBuilt *g) {
  const unsigned long long n = image->dev.bwtLength;
  /* 32-bit entries for the images that run 32-bit positions, 40-bit ones (DenseSa) for the others */
  const bool wide = !awfmImageNarrow(image);
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
        hipLaunchKernelGGL((packDense40Kernel<unsigned>), dim3((unsigned)image->numCUs * 8u), dim3(256), 0, awfmGpuSetupStream, (const unsigned *)stash, n, (unsigned *)other);
      } else {
        DenseSa from;
        from.words = (const unsigned *)stash;
        from.wide = 1u;
        hipLaunchKernelGGL(unpackDense40Kernel, dim3((unsigned)image->numCUs * 8u), dim3(256), 0, awfmGpuSetupStream, from, n, (unsigned *)other);
      }
      const bool ok = hipGetLastError() == hipSuccess && awfmGpuSetupSync() == hipSuccess;
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
  if (wide) return buildDenseSaWide(image, capped, g);
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
  unsigned stepCap = 32u * image->dev.saRatio;
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
    if (awfmGpuSetupMemset(counter, 0, 16) != hipSuccess) rc = AwFmGeneralFailure;
    for (unsigned long long first = 0; first < n && rc == AwFmSuccess; first += chunk) {
      const unsigned long long count = n - first < chunk ? n - first : chunk;
      hipLaunchKernelGGL(iotaKernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, awfmGpuSetupStream, chunkBuf, first, count);
      rc = awfmGpuLaunchLocate(image, count, chunkBuf, awfmGpuSetupStream, nullptr, nullptr, stepCap);
      if (stepCap && listing)
        hipLaunchKernelGGL(narrowParkListKernel, dim3((unsigned)image->numCUs * 8u), dim3(256), 0, awfmGpuSetupStream, (const unsigned long long *)chunkBuf,
                           count, first, dense, listAt, listEntry, listCapacity, counter);
      else if (stepCap)
        hipLaunchKernelGGL(narrowParkKernel, dim3((unsigned)image->numCUs * 8u), dim3(256), 0, awfmGpuSetupStream, (const unsigned long long *)chunkBuf, count,
                           dense + first, park ? park + first : (unsigned long long *)nullptr, counter);
      else
        hipLaunchKernelGGL(narrowKernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, awfmGpuSetupStream, chunkBuf, count, dense + first);
      if (hipGetLastError() != hipSuccess) rc = AwFmGeneralFailure;
    }
    if (awfmGpuSetupSync() != hipSuccess) rc = AwFmGeneralFailure;
  };
  walkAll();
  bool jumping = rc == AwFmSuccess;
  if (jumping) {
    unsigned long long parked = 0, left = 0;
    if (awfmGpuSetupToHost(&parked, counter, 8) != hipSuccess) rc = AwFmGeneralFailure;
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
        if (rc == AwFmSuccess && awfmGpuSetupToHost(&parked, counter, 8) != hipSuccess) rc = AwFmGeneralFailure;
      }
    }
    left = jumping ? parked : 0;
    unsigned rounds = 0;
    for (; rc == AwFmSuccess && left != 0 && rounds < 64u; rounds++) {
      if (awfmGpuSetupMemset(counter + 1, 0, 8) != hipSuccess) rc = AwFmGeneralFailure;
      if (listed)
        hipLaunchKernelGGL(denseSaJumpListKernel, dim3((unsigned)image->numCUs * 8u), dim3(256), 0, awfmGpuSetupStream, dense, (const unsigned *)listAt, listEntry,
                           parked, n, counter + 1);
      else
        hipLaunchKernelGGL(denseSaJumpKernel, dim3((unsigned)image->numCUs * 8u), dim3(256), 0, awfmGpuSetupStream, dense, park, n, counter + 1);
      if (hipGetLastError() != hipSuccess || awfmGpuSetupToHost(&left, counter + 1, 8) != hipSuccess) rc = AwFmGeneralFailure;
    }
    if (awfmKnob(AWFM_KNOB_VERBOSE) && parked)
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
static enum AwFmReturnCode buildDenseSaWide(AwFmGpuIndex *image, bool capped, DenseBuilt *g) {
  const unsigned long long n = image->dev.bwtLength;
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
  unsigned stepCap = 32u * image->dev.saRatio;
  auto walkAll = [&]() {
    if (awfmGpuSetupMemset(counter, 0, 16) != hipSuccess) rc = AwFmGeneralFailure;
    for (unsigned long long first = 0; first < n && rc == AwFmSuccess; first += chunk) {
      const unsigned long long count = n - first < chunk ? n - first : chunk;
      hipLaunchKernelGGL(iotaKernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, awfmGpuSetupStream, chunkBuf, first, count);
      rc = awfmGpuLaunchLocate(image, count, chunkBuf, awfmGpuSetupStream, nullptr, nullptr, stepCap);
      if (stepCap)
        hipLaunchKernelGGL(parkWideKernel, dim3((unsigned)image->numCUs * 8u), dim3(256), 0, awfmGpuSetupStream, (const unsigned long long *)chunkBuf, count, first, dense,
                           listAt, entry[0], listCapacity, counter);
      else if (hipMemcpyAsync(dense + first, chunkBuf, count * 8, hipMemcpyDeviceToDevice, awfmGpuSetupStream) != hipSuccess)
        rc = AwFmGeneralFailure;
      if (hipGetLastError() != hipSuccess) rc = AwFmGeneralFailure;
    }
    if (awfmGpuSetupSync() != hipSuccess) rc = AwFmGeneralFailure;
  };
  walkAll();
  unsigned long long parked = 0, left = 0;
  if (rc == AwFmSuccess && awfmGpuSetupToHost(&parked, counter, 8) != hipSuccess) rc = AwFmGeneralFailure;
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
    if (awfmGpuSetupMemset(counter + 1, 0, 8) != hipSuccess) rc = AwFmGeneralFailure;
    hipLaunchKernelGGL(denseSaJumpWideKernel, dim3((unsigned)image->numCUs * 8u), dim3(256), 0, awfmGpuSetupStream, dense, (const unsigned long long *)listAt,
                       (const ulonglong2 *)entry[rounds & 1u], entry[(rounds & 1u) ^ 1u], parked, n, counter + 1);
    if (hipGetLastError() != hipSuccess || awfmGpuSetupToHost(&left, counter + 1, 8) != hipSuccess) rc = AwFmGeneralFailure;
  }
  if (awfmKnob(AWFM_KNOB_VERBOSE) && parked)
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
    hipLaunchKernelGGL((packDense40Kernel<unsigned long long>), dim3((unsigned)image->numCUs * 8u), dim3(256), 0, awfmGpuSetupStream, (const unsigned long long *)dense, n, packed);
    if (hipGetLastError() != hipSuccess || awfmGpuSetupSync() != hipSuccess) rc = AwFmGeneralFailure;
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
}  // extern "C"
enum AwFmReturnCode awfmGpuBuildDenseSaAuto(const AwFmGpuIndex *g, void **arrayOut, bool *wideOut, uint64_t *bytesOut, double *secondsOut,
                                            std::string *notes) {
  *arrayOut = nullptr;
  *wideOut = false;
  *bytesOut = 0;
  *secondsOut = 0.0;
  bool want = false, automatic = false;
  const char *env = awfmKnob(AWFM_KNOB_DENSE_SA);
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
    if (!want) *notes += "full suffix array: not built (less than 4 x its size free); ";
    automatic = true;
  }
  if (!want || g->dev.bwtLength >= (1ull << 40)) return AwFmSuccess;
  DeviceGuard guard(g->device);
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  DenseBuilt built;
  const enum AwFmReturnCode rc = buildDenseSa(const_cast<AwFmGpuIndex *>(g), automatic, &built); /* (reads the image only) */
  if (!built.dDenseSa) *notes += "full suffix array: not built (no device memory, or walks that could not be completed); ";
  clock_gettime(CLOCK_MONOTONIC, &t1);
  *arrayOut = built.dDenseSa;
  *wideOut = built.denseWide;
  *bytesOut = built.denseSaBytes;
  *secondsOut = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  return rc;
}
enum AwFmReturnCode awfmGpuApplyDenseSaAuto(AwFmGpuIndex *g) {
  void *array = nullptr;
  bool wide = false;
  uint64_t bytes = 0;
  double seconds = 0.0;
  const enum AwFmReturnCode rc = awfmGpuBuildDenseSaAuto(g, &array, &wide, &bytes, &seconds, &g->accelNotes);
  if (array) {
    if (g->dDenseSa) (void)hipFree(g->dDenseSa);
    g->dDenseSa = array;
    g->denseWide = wide;
    g->denseSaBytes = bytes;
    g->denseSaBuildSeconds = seconds;
  }
  return rc;
}
extern "C" {
int awfmGpuIndexHasDenseSa(const AwFmGpuIndex *g) { return g && g->dDenseSa ? 1 : 0; }
double awfmGpuIndexDenseSaBuildSeconds(const AwFmGpuIndex *g) { return g ? (g->shares ? g->shares : g)->denseSaBuildSeconds : 0.0; }
}  // extern "C"
