/*
 * awfm_locate_kernel.h -- suffix-array backtrace: BWT position of a hit -> text position.
 *
 * One hit is owned by a group of G lanes (same piece ownership as the search kernel: lane j holds
 * pieces j*S..j*S+S-1 of a block, S = 8/G).  Per LF step the group reads the block of the current
 * BWT position as whole 128-B lines, extracts the letter stored there (the owning lane builds the
 * code from its plane words, the group gets it by ds_bpermute), ranks that letter up to the position
 * and continues at C[a] + Occ(a,p) - 1 until the position is sampled; then one lane reads the
 * bit-packed sample and stores (SA_s[p/ratio] + steps) mod bwtLength in place.
 * ref src/AwFmParallelSearch.c:338-361, src/AwFmSearch.c:369-427, src/AwFmOccurrence.c:170-217,
 *     src/AwFmIndexStruct.c:88-91, src/AwFmSuffixArray.c:22-39, :114-142, :179-191.
 */
#ifndef AWFM_LOCATE_KERNEL_H
#define AWFM_LOCATE_KERNEL_H

#include "awfm_search_kernel.h"

namespace {

/* sampled SA value i from the little-endian bit stream */
__device__ __forceinline__ unsigned long long sampledSaValue(const DevIndex &ix, unsigned long long i) {
  const unsigned long long bit = i * ix.saWidth; /* bwtLength*width < 2^64 for any index that fits memory */
  const unsigned long long word = bit >> 6;
  const unsigned shift = (unsigned)bit & 63u;
  const unsigned long long lo = ix.sa[word];
  unsigned long long v = lo >> shift;
  if (shift + ix.saWidth > 64u) v |= ix.sa[word + 1ull] << (64u - shift);
  return ix.saWidth >= 64u ? v : v & ((1ull << ix.saWidth) - 1ull);
}

template <bool AMINO, int G>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_num_sgpr(80)))
    locateKernel(const DevIndex ix, unsigned long long totalHits, unsigned long long *__restrict__ positions) {
  constexpr int S = 8 / G;
  constexpr int V = AMINO ? 2 : 1;
  constexpr int kGroups = kThreads / G;
  __shared__ unsigned long long sC[24];
  __shared__ AminoShared sAmino;
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  if (AMINO && threadIdx.x < 32) {
    sAmino.letterOfAscii[threadIdx.x] = kAminoTables.letterOfAscii[threadIdx.x];
    sAmino.letterOfCode[threadIdx.x] = kAminoTables.letterOfCode[threadIdx.x];
    if (threadIdx.x < 24) sAmino.planeMask[threadIdx.x] = kAminoTables.planeMask[threadIdx.x];
  }
  __syncthreads();
  const unsigned gl = threadIdx.x % G;
  const unsigned firstPiece = gl * S;
  const unsigned long long numGroups = (unsigned long long)gridDim.x * kGroups;
  const bool pow2 = ix.saShift != 0xFFFFFFFFu;
  const unsigned long long ratio = ix.saRatio;

  /* Persistent walk with refill: every iteration each group performs exactly one dependent memory
   * round trip -- the block of its current BWT position (LF step) or, once the position is sampled,
   * the SA sample -- and a group that finishes a hit continues with its next one in the same
   * iteration.  (Chain lengths are roughly geometric with mean ratio-1; waiting for the longest of
   * the 64/G chains of a wave would leave most lanes idle.)  Both request kinds are issued before
   * either is consumed, so they overlap across the groups of a wave. */
  unsigned long long t = ((unsigned long long)blockIdx.x * kThreads + threadIdx.x) / G;
  bool alive = t < totalHits;
  unsigned long long p = alive ? positions[t] : 0ull;
  unsigned long long nextP = t + numGroups < totalHits ? positions[t + numGroups] : 0ull;
  unsigned long long offset = 0;
  while (alive) {
    const bool sampled = (pow2 ? (p & (ratio - 1ull)) == 0ull : (p % ratio) == 0ull) || offset > ix.bwtLength;
    const unsigned long long blk = p >> 8;
    const unsigned local = (unsigned)p & 255u;
    /* ---- issue ---- */
    uint4 pc[S][V];
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
      for (int v = 0; v < V; v++) pc[s][v] = make_uint4(0u, 0u, 0u, 0u);
    if (!sampled) {
#pragma unroll
      for (int s = 0; s < S; s++)
#pragma unroll
        for (int v = 0; v < V; v++) pc[s][v] = ix.blocks[(blk * 8ull + firstPiece + s) * V + v];
    }
    const unsigned long long sample = pow2 ? p >> ix.saShift : p / ratio;
    const unsigned long long saBit = sample * ix.saWidth; /* bwtLength*width < 2^64 for any index that fits memory */
    unsigned long long saLo = 0, saHi = 0;
    if (sampled) { /* the image keeps 16 bytes of slack behind the packed SA, so word+1 is always readable */
      saLo = ix.sa[saBit >> 6];
      saHi = ix.sa[(saBit >> 6) + 1ull];
    }
    /* ---- consume ---- */
    if (!sampled) {
      const unsigned bit = local & 31u, ownerPiece = local >> 5;
      /* letter stored at p: code bits from the owning piece */
      unsigned myCode = 0;
#pragma unroll
      for (int s = 0; s < S; s++) {
        unsigned code = ((pc[s][0].x >> bit) & 1u) | (((pc[s][0].y >> bit) & 1u) << 1) | (((pc[s][0].z >> bit) & 1u) << 2);
        if (AMINO) code |= (((pc[s][0].w >> bit) & 1u) << 3) | (((pc[s][V - 1].x >> bit) & 1u) << 4);
        myCode = (ownerPiece % S) == (unsigned)s ? code : myCode;
      }
      const unsigned code = groupShfl<G>(myCode, ownerPiece / S);
      unsigned long long next;
      if (AMINO) {
        const unsigned letter = sAmino.letterOfCode[code];
        const unsigned safe = letter < 21u ? letter : 0u; /* sentinel handled below */
        const unsigned pm = sAmino.planeMask[safe];
        const unsigned ones = pm & 0xFFu, zeros = pm >> 8;
        unsigned n = 0, mine = 0;
        const unsigned piece = safe / 3u, slot = safe % 3u;
#pragma unroll
        for (int s = 0; s < S; s++) {
          n += __popc(aminoOccSlice(pc[s][0], pc[s][V - 1], ones, zeros) & sliceMask(local, firstPiece + s));
          mine = (piece % S) == (unsigned)s ? aminoCountWord(pc[s][V - 1], slot) : mine;
        }
        next = sC[safe] + groupShfl<G>(mine, piece / S) + groupSum<G>(n) - 1ull;
        if (letter == 21u) next = 0; /* sentinel: ref src/AwFmSearch.c:414-416 */
      } else {
        const unsigned letter = (0x00152435u >> (4u * code)) & 7u; /* code -> index {5,3,4,2,5,1,0,0}, ref src/AwFmLetter.c:49-53 */
        const unsigned safe = letter < 5u ? letter : 0u;
        const PlaneSel3 sel = nucPlaneSel(safe);
        unsigned n = 0;
#pragma unroll
        for (int s = 0; s < S; s++) n += __popc(nucOccSlice(pc[s][0], sel) & sliceMask(local, firstPiece + s));
        unsigned long long base;
        if (safe < 4u) {
          const unsigned kLo = 2u * safe, kHi = kLo + 1u;
          unsigned lo = 0, hi = 0;
#pragma unroll
          for (int s = 0; s < S; s++) {
            lo = (kLo % S) == (unsigned)s ? pc[s][0].w : lo;
            hi = (kHi % S) == (unsigned)s ? pc[s][0].w : hi;
          }
          base = ((unsigned long long)groupShfl<G>(hi, kHi / S) << 32) | groupShfl<G>(lo, kLo / S);
        } else {
          unsigned long long part = 0;
#pragma unroll
          for (int s = 0; s < S; s++)
            part += ((firstPiece + s) & 1u) ? ((unsigned long long)pc[s][0].w << 32) : (unsigned long long)pc[s][0].w;
          const unsigned long long before = blk * 256ull;
          base = before - groupSum64<G>(part) - (ix.sentinelPos < before ? 1ull : 0ull);
        }
        next = sC[safe] + base + groupSum<G>(n) - 1ull;
        if (letter == 5u) next = 0; /* sentinel: ref src/AwFmSearch.c:384-386 */
      }
      p = next;
      offset++;
    } else {
      const unsigned shift = (unsigned)saBit & 63u;
      unsigned long long v = saLo >> shift;
      if (shift + ix.saWidth > 64u) v |= saHi << (64u - shift);
      if (ix.saWidth < 64u) v &= (1ull << ix.saWidth) - 1ull;
      v += offset;
      if (v >= ix.bwtLength) v -= ix.bwtLength;
      if (v >= ix.bwtLength) v %= ix.bwtLength;
      if (gl == 0) positions[t] = v;
      /* next hit of this group */
      t += numGroups;
      alive = t < totalHits;
      p = nextP;
      offset = 0;
      if (t + numGroups < totalHits) nextP = positions[t + numGroups];
    }
  }
}

}  // namespace

#endif
