/*
 * awfm_gpu_pair.hip -- construction of the pair image (awfm_pair.h) from the one-letter device image.
 *
 *   pairCodesKernel   one lane per BWT position p: letter at p, q = LF(p) (rank in p's own block), letter at q (one
 *                     random 16-B read), the pair code; a wave turns 64 positions into two slices of bit planes with
 *                     ballots and leaves a 32-byte histogram of its half block (16 pairs, 4 letters) for the counting pass
 *   pairCountsKernel  one workgroup per superblock of 2^24 positions: exclusive scan of the half-block histograms,
 *                     24-bit relative counts (pairs and letters) into the blocks, totals per superblock
 *   pairStartKernel   C2[c1c2] = C[c1] + Occ(c1, C[c2]) by one rank on the one-letter image each
 * and a 64-bit prefix sum over the few hundred superblock totals on the host.
 *
 * Everything is derived from the image (ref layout semantics: src/AwFmSearch.c:42-103, :369-427,
 * src/AwFmIndexStruct.c:88-91); the host index and its file are untouched.
 */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "awfm_pair.h"

namespace {

typedef unsigned long long u64;

/* occurrences of letter a (0..3) in L[0..q], absolute: one thread reads the whole block */
__device__ u64 nucRankScalar(const DevIndex &ix, unsigned a, u64 q) {
  const u64 blk = q >> kBlockShift;
  const unsigned local = (unsigned)q & kBlockMask;
  const PlaneSel3 sel = nucPlaneSel(a);
  u64 count = ix.super[(blk >> (ix.nucSuperShift - kBlockShift)) * 4ull + a];
  for (unsigned s = 0; s < kSlices; s++) {
    const uint4 raw = ix.blocks[blk * kSlices + s];
    Piece pc;
    pc.x = raw.x;
    pc.y = raw.y;
    pc.z = raw.z;
    pc.w = raw.w;
    if (s == a) count += pc.w;
    count += __popc(nucOccSlice(pc, sel) & sliceMask(local, s));
  }
  return count;
}

/* letter index (0..3 a,c,g,t; 4 x; 5 '$') stored at BWT position q */
__device__ unsigned nucLetterAt(const DevIndex &ix, u64 q) {
  const uint4 raw = ix.blocks[(q >> kBlockShift) * kSlices + (((unsigned)q & kBlockMask) >> 5)];
  const unsigned bit = (unsigned)q & 31u;
  const unsigned code = ((raw.x >> bit) & 1u) | (((raw.y >> bit) & 1u) << 1) | (((raw.z >> bit) & 1u) << 2);
  return (0x00152435u >> (4u * code)) & 7u; /* ref src/AwFmLetter.c:49-53 */
}

__global__ void __launch_bounds__(256)
    pairCodesKernel(const DevIndex ix, uint4 *__restrict__ pairBlocks, uint4 *__restrict__ halfHist) {
  __shared__ u64 sC[8];
  if (threadIdx.x < 6) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  __syncthreads();
  const u64 numHalves = ((ix.bwtLength + kBlockMask) >> kBlockShift) * 2ull;
  const unsigned lane = threadIdx.x & 63u;
  const u64 waveStride = (u64)gridDim.x * 4ull;
  for (u64 half = (u64)blockIdx.x * 4ull + (threadIdx.x >> 6); half < numHalves; half += waveStride) {
    const u64 p = half * 64ull + lane;
    bool live = p < ix.bwtLength, valid = false;
    unsigned pi = 0, letter = 7u;
    if (live) {
      const unsigned a = nucLetterAt(ix, p);
      letter = a;
      if (a < 4u) {
        const u64 q = sC[a] + nucRankScalar(ix, a, p) - 1ull; /* LF(p) */
        const unsigned a2 = nucLetterAt(ix, q);
        valid = a2 < 4u;
        pi = (a2 & 3u) * 4u + a;
      }
    }
    const u64 b0 = __ballot(pi & 1u), b1 = __ballot(pi & 2u), b2 = __ballot(pi & 4u), b3 = __ballot(pi & 8u);
    const bool flagged = __ballot(live && !valid) != 0ull;
    /* bytes 0..15 of the histogram: positions of this half block with pair 0..15 (at most 64 each); bit 7 of byte 0:
     * the half holds a position without a pair */
    unsigned mine = 0; /* lanes 0..15 each count one pair */
    for (unsigned i = 0; i < 16u; i++) {
      const unsigned n = (unsigned)__popcll(__ballot(valid && pi == i));
      mine = lane == i ? n : mine;
    }
    if (lane == 0 && flagged) mine |= 0x80u;
    /* bytes 16..19: positions of this half block whose own letter is a, c, g, t (whatever precedes it) */
    for (unsigned a = 0; a < 4u; a++) {
      const unsigned n = (unsigned)__popcll(__ballot(letter == a));
      mine = lane == 16u + a ? n : mine;
    }
    /* gather the 20 byte counts into lane 0 as two 16-byte words */
    unsigned w[5];
    for (unsigned j = 0; j < 5u; j++) {
      unsigned v = 0;
      for (unsigned i = 0; i < 4u; i++) v |= (unsigned)__shfl((int)mine, (int)(4u * j + i), 64) << (8u * i);
      w[j] = v;
    }
    if (lane == 0) {
      halfHist[2ull * half] = make_uint4(w[0], w[1], w[2], w[3]);
      halfHist[2ull * half + 1ull] = make_uint4(w[4], 0u, 0u, 0u);
    }
    /* the two slices of this half: planes piece by lanes 0 / 1 (the counts piece follows later) */
    if (lane < 2u) {
      const unsigned sh = 32u * lane;
      const u64 piece = pairPlanesAt(half >> 1, (unsigned)(half & 1ull) * 2u + lane);
      pairBlocks[piece] = make_uint4((unsigned)(b0 >> sh), (unsigned)(b1 >> sh), (unsigned)(b2 >> sh), (unsigned)(b3 >> sh));
    }
  }
}

/* one workgroup per superblock: thread t owns kPerThread consecutive blocks of it */
__global__ void __launch_bounds__(256)
    pairCountsKernel(const uint4 *__restrict__ halfHist, u64 numBlocks, uint4 *__restrict__ pairBlocks,
                     u64 *__restrict__ superTotals) {
  __shared__ unsigned sPart[256][kPairSuperStride + 1]; /* +1: no bank conflicts on the column walk */
  constexpr u64 kPerThread = (1ull << (kPairSuperShift - kBlockShift)) / 256ull;
  const u64 firstBlock = (u64)blockIdx.x << (kPairSuperShift - kBlockShift);
  const u64 myFirst = firstBlock + kPerThread * threadIdx.x;
  unsigned acc[kPairSuperStride]; /* 16 pairs, 4 letters */
  for (unsigned i = 0; i < kPairSuperStride; i++) acc[i] = 0;
  auto addHalf = [&](u64 half) -> unsigned { /* returns the flag bit of the half */
    const uint4 h = halfHist[2ull * half], g = halfHist[2ull * half + 1ull];
    const unsigned w[5] = {h.x & ~0x80u, h.y, h.z, h.w, g.x};
    for (unsigned i = 0; i < kPairSuperStride; i++) acc[i] += (w[i >> 2] >> (8u * (i & 3u))) & 0xFFu;
    return h.x & 0x80u;
  };
  for (u64 b = myFirst; b < myFirst + kPerThread && b < numBlocks; b++) {
    (void)addHalf(2ull * b);
    (void)addHalf(2ull * b + 1ull);
  }
  for (unsigned i = 0; i < kPairSuperStride; i++) sPart[threadIdx.x][i] = acc[i];
  __syncthreads();
  if (threadIdx.x < kPairSuperStride) { /* exclusive scan down each column */
    unsigned run = 0;
    for (int t = 0; t < 256; t++) {
      const unsigned v = sPart[t][threadIdx.x];
      sPart[t][threadIdx.x] = run;
      run += v;
    }
    superTotals[(u64)blockIdx.x * kPairSuperStride + threadIdx.x] = run;
  }
  __syncthreads();
  for (unsigned i = 0; i < kPairSuperStride; i++) acc[i] = sPart[threadIdx.x][i];
  for (u64 b = myFirst; b < myFirst + kPerThread && b < numBlocks; b++) {
    unsigned before[kPairSuperStride];
    for (unsigned i = 0; i < kPairSuperStride; i++) before[i] = acc[i];
    const unsigned flag = (addHalf(2ull * b) | addHalf(2ull * b + 1ull)) ? 0x80000000u : 0u;
    for (unsigned k = 0; k < 4u; k++) { /* slice k: count of letter k (+ the flag), counts of pairs 4k..4k+3, 24 bits each */
      const unsigned c0 = before[4 * k], c1 = before[4 * k + 1], c2 = before[4 * k + 2], c3 = before[4 * k + 3];
      pairBlocks[pairCountsAt(b, k)] =
          make_uint4(before[16u + k] | flag, c0 | (c1 << 24), (c1 >> 8) | (c2 << 16), (c2 >> 16) | (c3 << 8));
    }
  }
}

__global__ void pairStartKernel(const DevIndex ix, u64 *__restrict__ pairC) {
  const unsigned pi = threadIdx.x;
  if (pi >= 16u) return;
  const unsigned c1 = pi >> 2, c2 = pi & 3u;
  const u64 startC2 = ix.prefixSums[c2]; /* first row of the suffixes that start with c2 (>= 1: row 0 is "$") */
  pairC[pi] = ix.prefixSums[c1] + nucRankScalar(ix, c1, startC2 - 1ull);
}

}  // namespace

enum AwFmReturnCode awfmGpuApplyPairImage(AwFmGpuIndex *g, bool enable) {
  (void)hipDeviceSynchronize();
  void **owned[] = {&g->dPairBlocks, &g->dPairSuper, &g->dPairSuper32, &g->dPairC};
  /* pairSuper / pairSuper32: kPairSuperStride = 20 words per superblock (16 pairs, then the 4 letters) */
  for (void **p : owned) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  g->pairBytes = 0;
  g->dev.pairBlocks = nullptr;
  g->dev.pairSuper = nullptr;
  g->dev.pairSuper32 = nullptr;
  g->dev.pairC = nullptr;
  g->dev.numPairSuper = 0;
  if (!enable || g->amino) return AwFmSuccess;
  const u64 n = g->dev.bwtLength;
  const u64 numBlocks = (n + kBlockMask) >> kBlockShift;
  const u64 numSuper = ((n - 1ull) >> kPairSuperShift) + 1ull;
  if (numSuper > (1ull << 20)) {
    setError("pair image: more than 2^43 positions");
    return AwFmUnsupportedVersionError;
  }
  uint4 *blocks = nullptr, *hist = nullptr;
  u64 *totals = nullptr, *super = nullptr, *pairC = nullptr;
  unsigned *super32 = nullptr;
  auto fail = [&](const char *what, hipError_t e) {
    setError(what, e);
    void *all[] = {blocks, hist, totals, super, pairC, super32};
    for (void *p : all)
      if (p) (void)hipFree(p);
    return AwFmAllocationFailure;
  };
  hipError_t e;
  if ((e = hipMalloc((void **)&blocks, numBlocks * 128ull)) != hipSuccess) return fail("pair image: blocks", e);
  if ((e = hipMalloc((void **)&hist, numBlocks * 64ull)) != hipSuccess) return fail("pair image: histogram scratch", e);
  if ((e = hipMalloc((void **)&totals, numSuper * kPairSuperStride * 8ull)) != hipSuccess) return fail("pair image: totals", e);
  if ((e = hipMalloc((void **)&super, numSuper * kPairSuperStride * 8ull)) != hipSuccess) return fail("pair image: superblocks", e);
  if ((e = hipMalloc((void **)&super32, numSuper * kPairSuperStride * 4ull)) != hipSuccess) return fail("pair image: superblocks (32-bit)", e);
  if ((e = hipMalloc((void **)&pairC, 128)) != hipSuccess) return fail("pair image: starts", e);
  const u64 halves = numBlocks * 2ull;
  const unsigned codesGrid = (unsigned)((halves + 3ull) / 4ull < (u64)g->numCUs * 32ull ? (halves + 3ull) / 4ull : (u64)g->numCUs * 32ull);
  hipLaunchKernelGGL(pairCodesKernel, dim3(codesGrid ? codesGrid : 1u), dim3(256), 0, 0, g->dev, blocks, hist);
  /* (the letters' counts include the positions of flagged blocks: they are counted from L itself) */
  hipLaunchKernelGGL(pairCountsKernel, dim3((unsigned)numSuper), dim3(256), 0, 0, (const uint4 *)hist, numBlocks, blocks, totals);
  hipLaunchKernelGGL(pairStartKernel, dim3(1), dim3(64), 0, 0, g->dev, pairC);
  if ((e = hipGetLastError()) != hipSuccess) return fail("pair image: launch", e);
  const u64 words = numSuper * kPairSuperStride;
  std::vector<u64> host(words), bases(words);
  std::vector<unsigned> bases32(words);
  if ((e = hipMemcpy(host.data(), totals, words * 8ull, hipMemcpyDeviceToHost)) != hipSuccess) return fail("pair image: totals", e);
  for (unsigned pi = 0; pi < kPairSuperStride; pi++) {
    u64 run = 0;
    for (u64 sb = 0; sb < numSuper; sb++) {
      bases[sb * kPairSuperStride + pi] = run;
      bases32[sb * kPairSuperStride + pi] = (unsigned)run; /* exact below 2^32 positions, the only images that read it */
      run += host[sb * kPairSuperStride + pi];
    }
  }
  if ((e = hipMemcpy(super, bases.data(), words * 8ull, hipMemcpyHostToDevice)) != hipSuccess) return fail("pair image: bases", e);
  if ((e = hipMemcpy(super32, bases32.data(), words * 4ull, hipMemcpyHostToDevice)) != hipSuccess) return fail("pair image: bases", e);
  (void)hipFree(hist);
  (void)hipFree(totals);
  g->dPairBlocks = blocks;
  g->dPairSuper = super;
  g->dPairSuper32 = super32;
  g->dPairC = pairC;
  g->pairBytes = numBlocks * 128ull + numSuper * kPairSuperStride * 12ull + 128ull;
  g->dev.pairBlocks = blocks;
  g->dev.pairSuper = super;
  g->dev.pairSuper32 = super32;
  g->dev.pairC = pairC;
  g->dev.numPairSuper = (unsigned)numSuper;
  return AwFmSuccess;
}
