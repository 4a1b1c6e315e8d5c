/*
 * awfm_gpu.hip -- HIP (gfx950 / CDNA4) side of libawfmindex_amd.so.
 *
 * Device image ("re-laid-out windowed BWT"):
 *   nucleotide block = 128 B = one HBM line = 8 pieces of 16 B; piece k holds,
 *     for BWT positions 32k..32k+31 of the block, the three plane words
 *     {b0,b1,b2} and one 32-bit word of the A/C/G/T base counts
 *     (word 2a = low half, 2a+1 = high half of count[a]).  The X count is
 *     derived: positions before the block minus A+C+G+T minus the sentinel.
 *   amino block = 256 B = 8 pieces of 32 B; piece k holds the five plane words
 *     and three 32-bit base counts (slot 3k+s = letter 3k+s, 21 letters incl. Z);
 *     needs bwtLength < 2^32.
 *   The reference layout (ref src/AwFmIndex.h:55-65: 160 / 352 B blocks) always
 *   straddles two 128-B lines per rank; here a rank reads exactly 1 (2) lines.
 *
 * Search kernel ("group8"): 8 lanes cooperate on one query; one
 * global_load_dwordx4 per lane fetches a whole block as one fully used 128-B
 * request, every lane ranks its own 32 positions with AND/XOR/popcount and the
 * eight partial counts are summed with DPP adds (no LDS traffic, no bank
 * conflicts).  A wave therefore has 8 queries x 2 blocks = up to 16 lines in
 * flight per step; occupancy (32 waves/CU) supplies the rest of the memory
 * level parallelism.
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

#include "awfm_internal.h"

namespace {

thread_local std::string tlsError;

void setError(const char *what, hipError_t e) {
  tlsError = std::string(what) + ": " + hipGetErrorString(e);
}
void setError(const char *what) { tlsError = what; }

#define AWFM_HIP_TRY(call, failRc)                      \
  do {                                                  \
    hipError_t err__ = (call);                          \
    if (err__ != hipSuccess) {                          \
      setError(#call, err__);                           \
      return (failRc);                                  \
    }                                                   \
  } while (0)

/* kernel-argument view of the device image */
struct DevIndex {
  const uint4 *blocks;
  const ulonglong2 *seed;
  const unsigned long long *sa; /* packed sampled SA viewed as 64-bit words */
  unsigned long long bwtLength;
  unsigned long long sentinelPos; /* BWT position holding '$' */
  unsigned long long seedLen;
  const unsigned long long *prefixSums; /* 24 words in device memory */
  unsigned int saRatio;
  unsigned int saShift; /* log2(saRatio) when it is a power of two, else 0xFFFFFFFF */
  unsigned int saWidth;
  unsigned int seedK;
};

constexpr int kThreads = 256;
constexpr int kGroupsPerBlock = kThreads / 8;

/* ------------------------------------------------------------------ device helpers */

template <int CTRL>
__device__ __forceinline__ unsigned dppMove(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

/* sum over the 8 lanes of a group; every lane gets the total */
__device__ __forceinline__ unsigned groupSum8(unsigned v) {
  v += dppMove<0xB1>(v);  /* quad_perm [1,0,3,2] */
  v += dppMove<0x4E>(v);  /* quad_perm [2,3,0,1] */
  v += dppMove<0x141>(v); /* row_half_mirror: lane i <- lane 7-i of its half row */
  return v;
}

__device__ __forceinline__ unsigned long long groupSum8u64(unsigned long long v) {
#pragma unroll
  for (int stage = 0; stage < 3; stage++) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32), olo, ohi;
    if (stage == 0) {
      olo = dppMove<0xB1>(lo);
      ohi = dppMove<0xB1>(hi);
    } else if (stage == 1) {
      olo = dppMove<0x4E>(lo);
      ohi = dppMove<0x4E>(hi);
    } else {
      olo = dppMove<0x141>(lo);
      ohi = dppMove<0x141>(hi);
    }
    v += ((unsigned long long)ohi << 32) | olo;
  }
  return v;
}

/* bits 0..(p - 32*piece) of a 32-position slice, clamped: the slice's share of
 * the inclusive prefix mask of ref src/AwFmSimdConfig.c:89-114 */
__device__ __forceinline__ unsigned sliceMask(unsigned p, unsigned piece) {
  int bits = (int)p - (int)(piece * 32) + 1;
  bits = bits < 0 ? 0 : (bits > 32 ? 32 : bits);
  return (unsigned)((1ull << bits) - 1ull);
}

/* ---- nucleotide ---- */

/* ref src/AwFmLetter.c:4-22 */
__device__ __forceinline__ unsigned nucLetterIndex(unsigned c) {
  const unsigned l = c | 0x20u;
  return l == 'a' ? 0u : l == 'c' ? 1u : l == 'g' ? 2u : (l == 't' || l == 'u') ? 3u : l == '$' ? 5u : 4u;
}
/* ref src/AwFmLetter.c:98-125 */
__device__ __forceinline__ bool nucIsAmbiguous(unsigned c) {
  const unsigned l = (c >= 'A' && c <= 'Z') ? (c | 0x20u) : c;
  return !(l == 'a' || l == 'c' || l == 'g' || l == 't' || l == 'u');
}

struct PlaneSel3 {
  unsigned x0, x1, x2; /* all-ones where the plane must be 0 */
  unsigned d0, d1, d2; /* all-ones where the plane is don't-care */
};

/* plane literals of ref src/AwFmOccurrence.c:18-31: ones {6,5,3,1,2}, zeros {0,0,0,6,5} */
__device__ __forceinline__ PlaneSel3 nucPlaneSel(unsigned letter) {
  const unsigned ones = (0x21356u >> (4u * letter)) & 7u;
  const unsigned zeros = (0x56000u >> (4u * letter)) & 7u;
  const unsigned care = ones | zeros;
  PlaneSel3 s;
  s.x0 = 0u - (zeros & 1u);
  s.x1 = 0u - ((zeros >> 1) & 1u);
  s.x2 = 0u - ((zeros >> 2) & 1u);
  s.d0 = (care & 1u) - 1u;
  s.d1 = ((care >> 1) & 1u) - 1u;
  s.d2 = ((care >> 2) & 1u) - 1u;
  return s;
}

__device__ __forceinline__ unsigned nucOccSlice(const uint4 &pc, const PlaneSel3 &s) {
  return ((pc.x ^ s.x0) | s.d0) & ((pc.y ^ s.x1) | s.d1) & ((pc.z ^ s.x2) | s.d2);
}

/* base count of `letter` before block `blk` from the count words spread over the group */
__device__ __forceinline__ unsigned long long nucBase(const uint4 &pc, unsigned letter, unsigned long long blk,
                                                      unsigned long long sentinelPos, unsigned g) {
  if (letter < 4u) {
    const unsigned lo = (unsigned)__shfl((int)pc.w, (int)(2u * letter), 8);
    const unsigned hi = (unsigned)__shfl((int)pc.w, (int)(2u * letter + 1u), 8);
    return ((unsigned long long)hi << 32) | lo;
  }
  /* X (or anything else): everything before the block that is not A,C,G,T,$ */
  const unsigned long long part = (g & 1u) ? ((unsigned long long)pc.w << 32) : (unsigned long long)pc.w;
  const unsigned long long acgt = groupSum8u64(part);
  const unsigned long long before = blk * 256ull;
  return before - acgt - (sentinelPos < before ? 1ull : 0ull);
}

/* one backward step for the group's query (ref src/AwFmSearch.c:42-103) */
__device__ __forceinline__ void nucStep(const DevIndex &ix, const unsigned long long *sC, unsigned letter,
                                        unsigned long long &sp, unsigned long long &ep, unsigned g) {
  const unsigned long long q0 = sp - 1ull, q1 = ep;
  const unsigned long long blk0 = q0 >> 8, blk1 = q1 >> 8;
  const uint4 pc0 = ix.blocks[blk0 * 8ull + g];
  uint4 pc1 = pc0;
  if (blk1 != blk0) pc1 = ix.blocks[blk1 * 8ull + g];
  const PlaneSel3 sel = nucPlaneSel(letter);
  const unsigned n0 = __popc(nucOccSlice(pc0, sel) & sliceMask((unsigned)q0 & 255u, g));
  const unsigned n1 = __popc(nucOccSlice(pc1, sel) & sliceMask((unsigned)q1 & 255u, g));
  const unsigned packed = groupSum8(n0 | (n1 << 16));
  const unsigned long long base0 = nucBase(pc0, letter, blk0, ix.sentinelPos, g);
  const unsigned long long base1 = nucBase(pc1, letter, blk1, ix.sentinelPos, g);
  const unsigned long long c = sC[letter];
  sp = c + base0 + (packed & 0xFFFFu);
  ep = c + base1 + (packed >> 16) - 1ull;
}

/* ---- amino ---- */

struct AminoTables {
  unsigned char letterOfAscii[32]; /* ref src/AwFmLetter.c:55-67 */
  unsigned char letterOfCode[32];  /* ref src/AwFmLetter.c:89-96 */
  unsigned short planeMask[24];    /* ones | zeros << 8, ref src/AwFmOccurrence.c:66-128 */
};

__constant__ AminoTables kAminoTables = {
    {20, 0,  20, 1,  2,  3,  4,  5,  6,  7,  20, 8,  9,  10, 11, 20,
     12, 13, 14, 15, 16, 20, 17, 18, 20, 19, 20, 20, 20, 20, 20, 20},
    {21, 18, 19, 2,  13, 16, 3,  20, 11, 12, 15, 20, 0, 20, 20, 20,
     20, 20, 20, 14, 20, 8,  17, 1,  20, 7,  5,  6,  9, 10, 4,  20},
    {0x0C | 0x10 << 8, 0x07 | 0x08 << 8, 0x03 | 0x10 << 8, 0x06 | 0x10 << 8, 0x0E | 0x01 << 8, 0x10 | 0x05 << 8,
     0x0B | 0x04 << 8, 0x10 | 0x06 << 8, 0x10 | 0x0A << 8, 0x10 | 0x03 << 8, 0x0D | 0x02 << 8, 0x08 | 0x07 << 8,
     0x09 | 0x10 << 8, 0x04 | 0x0B << 8, 0x10 | 0x0C << 8, 0x0A | 0x10 << 8, 0x05 | 0x10 << 8, 0x10 | 0x09 << 8,
     0x01 | 0x0E << 8, 0x02 | 0x0D << 8, 0x0F | 0x00 << 8, 0, 0, 0}};

struct AminoShared {
  unsigned char letterOfAscii[32];
  unsigned char letterOfCode[32];
  unsigned short planeMask[24];
};

__device__ __forceinline__ unsigned aminoLetterIndex(const AminoShared &t, unsigned c) {
  return c == '$' ? 21u : (unsigned)t.letterOfAscii[c & 31u];
}
__device__ __forceinline__ bool aminoIsAmbiguous(unsigned c) {
  const unsigned l = (c >= 'A' && c <= 'Z') ? (c | 0x20u) : c;
  return l == 'z' || l == 'x' || l == 'b';
}

/* An amino piece is two 16-B loads kept as plain uint4 values: lo = {b0,b1,b2,b3},
 * hi = {b4,c0,c1,c2} (plane words of this lane's 32 positions, then the base
 * counts of letters 3k, 3k+1, 3k+2). */
__device__ __forceinline__ unsigned aminoLiteral(unsigned plane, unsigned ones, unsigned zeros, unsigned j) {
  const unsigned x = 0u - ((zeros >> j) & 1u);
  const unsigned d = (((ones | zeros) >> j) & 1u) - 1u;
  return (plane ^ x) | d;
}

__device__ __forceinline__ unsigned aminoOccSlice(const uint4 &lo, const uint4 &hi, unsigned ones, unsigned zeros) {
  return aminoLiteral(lo.x, ones, zeros, 0) & aminoLiteral(lo.y, ones, zeros, 1) &
         aminoLiteral(lo.z, ones, zeros, 2) & aminoLiteral(lo.w, ones, zeros, 3) &
         aminoLiteral(hi.x, ones, zeros, 4);
}

__device__ __forceinline__ unsigned long long aminoBase(const uint4 &hi, unsigned letter) {
  /* pick count word 1+letter%3 of `hi` with shifts (a select chain on vector
   * components makes hipcc spill the vector to LDS for dynamic indexing) */
  const unsigned slot = letter % 3u;
  const unsigned long long c01 = ((unsigned long long)hi.z << 32) | hi.y;
  const unsigned long long c2x = hi.w;
  const unsigned mine = (unsigned)((slot == 2u ? c2x : c01) >> (slot == 1u ? 32u : 0u));
  return (unsigned)__shfl((int)mine, (int)(letter / 3u), 8);
}

/* ref src/AwFmSearch.c:105-159 */
__device__ __forceinline__ void aminoStep(const DevIndex &ix, const unsigned long long *sC, const AminoShared &t,
                                          unsigned letter, unsigned long long &sp, unsigned long long &ep,
                                          unsigned g) {
  const unsigned long long q0 = sp - 1ull, q1 = ep;
  const unsigned long long blk0 = q0 >> 8, blk1 = q1 >> 8;
  const uint4 lo0 = ix.blocks[blk0 * 16ull + 2u * g];
  const uint4 hi0 = ix.blocks[blk0 * 16ull + 2u * g + 1u];
  uint4 lo1 = lo0, hi1 = hi0;
  if (blk1 != blk0) {
    lo1 = ix.blocks[blk1 * 16ull + 2u * g];
    hi1 = ix.blocks[blk1 * 16ull + 2u * g + 1u];
  }
  const unsigned pm = t.planeMask[letter < 24u ? letter : 23u];
  const unsigned ones = pm & 0xFFu, zeros = pm >> 8;
  const unsigned n0 = __popc(aminoOccSlice(lo0, hi0, ones, zeros) & sliceMask((unsigned)q0 & 255u, g));
  const unsigned n1 = __popc(aminoOccSlice(lo1, hi1, ones, zeros) & sliceMask((unsigned)q1 & 255u, g));
  const unsigned packed = groupSum8(n0 | (n1 << 16));
  const unsigned long long c = sC[letter];
  sp = c + aminoBase(hi0, letter) + (packed & 0xFFFFu);
  ep = c + aminoBase(hi1, letter) + (packed >> 16) - 1ull;
}

/* ------------------------------------------------------------------ search kernel */

/* Seed + extend for one query per 8-lane group
 * (ref src/AwFmParallelSearch.c:222-313, src/AwFmKmerTable.c:4-51, src/AwFmSearch.c:485-520).
 * The non-seeded search over the last min(len,k) characters followed by the
 * extension loop is one right-to-left walk that stops at the first invalid range. */
template <bool AMINO>
__global__ void __launch_bounds__(kThreads)
    searchGroup8Kernel(const DevIndex ix, const unsigned char *__restrict__ chars,
                       const unsigned long long *__restrict__ offsets, const unsigned fixedLength,
                       const unsigned long long numQueries, ulonglong2 *__restrict__ ranges,
                       unsigned *__restrict__ counts) {
  __shared__ unsigned long long sC[24];
  __shared__ unsigned sPow[32];
  __shared__ AminoShared sAmino;
  const unsigned card = AMINO ? 20u : 4u;
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  if (threadIdx.x < 32) {
    /* weight of seed character i: card^(k-1-i) (ref src/AwFmKmerTable.c:26-32) */
    unsigned w = 1;
    for (unsigned e = threadIdx.x + 1; e < ix.seedK; e++) w *= card;
    sPow[threadIdx.x] = w;
    if (AMINO) {
      sAmino.letterOfAscii[threadIdx.x] = kAminoTables.letterOfAscii[threadIdx.x];
      sAmino.letterOfCode[threadIdx.x] = kAminoTables.letterOfCode[threadIdx.x];
      if (threadIdx.x < 24) sAmino.planeMask[threadIdx.x] = kAminoTables.planeMask[threadIdx.x];
    }
  }
  __syncthreads();

  const unsigned lane = threadIdx.x & 63u;
  const unsigned g = threadIdx.x & 7u;
  const unsigned long long numGroups = (unsigned long long)gridDim.x * kGroupsPerBlock;
  const unsigned K = ix.seedK;

  for (unsigned long long q = ((unsigned long long)blockIdx.x * kThreads + threadIdx.x) >> 3; q < numQueries;
       q += numGroups) {
    unsigned long long off, len;
    if (offsets) {
      off = offsets[q];
      len = offsets[q + 1] - off;
    } else {
      off = q * fixedLength;
      len = fixedLength;
    }
    const unsigned char *kmer = chars + off;
    unsigned long long sp = 1, ep = 0;
    long long pos = -1;
    if (len != 0) {
      bool seeded = false;
      if (K != 0 && len >= K) { /* ref src/AwFmKmerTable.c:4-19 */
        unsigned index = 0;
        bool ambiguous = false;
        for (unsigned i = g; i < K; i += 8) {
          const unsigned c = kmer[len - K + i];
          ambiguous |= AMINO ? aminoIsAmbiguous(c) : nucIsAmbiguous(c);
          index += (AMINO ? aminoLetterIndex(sAmino, c) : nucLetterIndex(c)) * sPow[i];
        }
        index = groupSum8(index);
        const unsigned long long ballot = __ballot(ambiguous);
        seeded = ((ballot >> (lane & 56u)) & 0xFFull) == 0ull;
        if (seeded && index < ix.seedLen) {
          const ulonglong2 r = ix.seed[index];
          sp = r.x;
          ep = r.y;
          pos = (long long)(len - K) - 1;
        } else {
          seeded = false;
        }
      }
      if (!seeded) { /* ref src/AwFmSearch.c:485-502 */
        const unsigned c = kmer[len - 1];
        const unsigned a = AMINO ? aminoLetterIndex(sAmino, c) : nucLetterIndex(c);
        sp = sC[a];
        ep = sC[a + 1] - 1ull;
        pos = (long long)len - 2;
      }
    }
    while (pos >= 0 && sp <= ep) {
      const unsigned c = kmer[pos];
      if (AMINO)
        aminoStep(ix, sC, sAmino, aminoLetterIndex(sAmino, c), sp, ep, g);
      else
        nucStep(ix, sC, nucLetterIndex(c), sp, ep, g);
      pos--;
    }
    if (g == 0) {
      if (ranges) ranges[q] = make_ulonglong2(sp, ep);
      if (counts) counts[q] = sp <= ep ? (unsigned)(ep - sp + 1ull) : 0u;
    }
  }
}

/* ------------------------------------------------------------------ locate kernels */

/* dLengths[i] = range length (ref src/AwFmIndexStruct.c:126-130) */
__global__ void rangeLengthKernel(const ulonglong2 *__restrict__ ranges, unsigned long long n,
                                  unsigned long long *__restrict__ lengths) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const ulonglong2 r = ranges[i];
    lengths[i] = r.x <= r.y ? r.y - r.x + 1ull : 0ull;
  }
}

constexpr int kScanThreads = 256;
constexpr int kScanItems = 4;
constexpr int kScanTile = kScanThreads * kScanItems;

/* per-tile sums */
__global__ void __launch_bounds__(kScanThreads)
    scanReduceKernel(const unsigned long long *__restrict__ in, unsigned long long n,
                     unsigned long long *__restrict__ tileSums) {
  __shared__ unsigned long long sWave[kScanThreads / 64];
  const unsigned long long base = (unsigned long long)blockIdx.x * kScanTile;
  unsigned long long v = 0;
  for (int k = 0; k < kScanItems; k++) {
    const unsigned long long i = base + (unsigned long long)k * kScanThreads + threadIdx.x;
    if (i < n) v += in[i];
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
__global__ void __launch_bounds__(kScanThreads)
    scanTileKernel(const unsigned long long *__restrict__ in, unsigned long long n,
                   const unsigned long long *__restrict__ tileOffsets, unsigned long long *__restrict__ out,
                   int writeTotal) {
  __shared__ unsigned long long sWave[kScanThreads / 64];
  const unsigned long long base = (unsigned long long)blockIdx.x * kScanTile + (unsigned long long)threadIdx.x * kScanItems;
  unsigned long long vals[kScanItems];
  unsigned long long sum = 0;
  for (int k = 0; k < kScanItems; k++) {
    vals[k] = base + k < n ? in[base + k] : 0ull;
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

/* dPositions[hitOffsets[i] + h] = sp_i + h (the BWT positions to trace back) */
__global__ void expandHitsKernel(const ulonglong2 *__restrict__ ranges, const unsigned long long *__restrict__ hitOffsets,
                                 unsigned long long n, unsigned long long *__restrict__ positions) {
  /* one wave per 64 queries: short lists by their own lane, long lists by the whole wave */
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long start = 0, count = 0, sp = 0;
  if (i < n) {
    const ulonglong2 r = ranges[i];
    sp = r.x;
    start = hitOffsets[i];
    count = hitOffsets[i + 1] - start;
  }
  const bool isLong = count > 32ull;
  if (!isLong)
    for (unsigned long long h = 0; h < count; h++) positions[start + h] = sp + h;
  unsigned long long longMask = __ballot(isLong);
  const unsigned lane = threadIdx.x & 63u;
  while (longMask) {
    const int src = __ffsll((long long)longMask) - 1;
    longMask &= longMask - 1ull;
    const unsigned long long s = __shfl(start, src, 64), c = __shfl(count, src, 64), p = __shfl(sp, src, 64);
    for (unsigned long long h = lane; h < c; h += 64ull) positions[s + h] = p + h;
  }
}

/* sampled SA value i from the little-endian bit stream (ref src/AwFmSuffixArray.c:22-39, :114-142) */
__device__ __forceinline__ unsigned long long sampledSaValue(const DevIndex &ix, unsigned long long i) {
  const unsigned long long bit = i * ix.saWidth; /* bwtLength*width < 2^64 for any index that fits memory */
  const unsigned long long word = bit >> 6;
  const unsigned shift = (unsigned)bit & 63u;
  const unsigned long long lo = ix.sa[word];
  unsigned long long v = lo >> shift;
  if (shift + ix.saWidth > 64u) v |= ix.sa[word + 1ull] << (64u - shift);
  return ix.saWidth >= 64u ? v : v & ((1ull << ix.saWidth) - 1ull);
}

/* LF walk to a sampled position + SA read, one hit per 8-lane group, in place
 * (ref src/AwFmParallelSearch.c:338-361, src/AwFmSearch.c:369-427,
 *  src/AwFmOccurrence.c:170-217, src/AwFmSuffixArray.c:179-191) */
template <bool AMINO>
__global__ void __launch_bounds__(kThreads)
    locateGroup8Kernel(const DevIndex ix, unsigned long long totalHits, unsigned long long *__restrict__ positions) {
  __shared__ unsigned long long sC[24];
  __shared__ AminoShared sAmino;
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  if (AMINO && threadIdx.x < 32) {
    sAmino.letterOfAscii[threadIdx.x] = kAminoTables.letterOfAscii[threadIdx.x];
    sAmino.letterOfCode[threadIdx.x] = kAminoTables.letterOfCode[threadIdx.x];
    if (threadIdx.x < 24) sAmino.planeMask[threadIdx.x] = kAminoTables.planeMask[threadIdx.x];
  }
  __syncthreads();
  const unsigned g = threadIdx.x & 7u;
  const unsigned long long numGroups = (unsigned long long)gridDim.x * kGroupsPerBlock;
  const bool pow2 = ix.saShift != 0xFFFFFFFFu;
  const unsigned long long ratio = ix.saRatio;

  for (unsigned long long t = ((unsigned long long)blockIdx.x * kThreads + threadIdx.x) >> 3; t < totalHits;
       t += numGroups) {
    unsigned long long p = positions[t];
    unsigned long long offset = 0;
    for (;;) {
      const bool sampled = pow2 ? (p & (ratio - 1ull)) == 0ull : (p % ratio) == 0ull; /* ref src/AwFmIndexStruct.c:88-91 */
      if (sampled || offset > ix.bwtLength) break; /* the bound only trips on a corrupt index */
      const unsigned long long blk = p >> 8;
      const unsigned local = (unsigned)p & 255u;
      const unsigned bit = local & 31u, owner = local >> 5;
      unsigned letter;
      unsigned long long next;
      if (AMINO) {
        const uint4 lo = ix.blocks[blk * 16ull + 2u * g];
        const uint4 hi = ix.blocks[blk * 16ull + 2u * g + 1u];
        const unsigned myCode = ((lo.x >> bit) & 1u) | (((lo.y >> bit) & 1u) << 1) | (((lo.z >> bit) & 1u) << 2) |
                                (((lo.w >> bit) & 1u) << 3) | (((hi.x >> bit) & 1u) << 4);
        const unsigned code = (unsigned)__shfl((int)myCode, (int)owner, 8);
        letter = sAmino.letterOfCode[code];
        if (letter == 21u) {
          next = 0;
        } else {
          const unsigned pm = sAmino.planeMask[letter];
          const unsigned n = groupSum8(__popc(aminoOccSlice(lo, hi, pm & 0xFFu, pm >> 8) & sliceMask(local, g)));
          next = sC[letter] + aminoBase(hi, letter) + n - 1ull;
        }
      } else {
        const uint4 pc = ix.blocks[blk * 8ull + g];
        const unsigned myCode = ((pc.x >> bit) & 1u) | (((pc.y >> bit) & 1u) << 1) | (((pc.z >> bit) & 1u) << 2);
        const unsigned code = (unsigned)__shfl((int)myCode, (int)owner, 8);
        letter = (0x00152435u >> (4u * code)) & 7u; /* code -> index {5,3,4,2,5,1,0,0}, ref src/AwFmLetter.c:49-53 */
        if (letter == 5u) {
          next = 0;
        } else {
          const PlaneSel3 sel = nucPlaneSel(letter);
          const unsigned n = groupSum8(__popc(nucOccSlice(pc, sel) & sliceMask(local, g)));
          next = sC[letter] + nucBase(pc, letter, blk, ix.sentinelPos, g) + n - 1ull;
        }
      }
      p = next;
      offset++;
    }
    if (g == 0) {
      const unsigned long long sample = pow2 ? p >> ix.saShift : p / ratio;
      unsigned long long v = sampledSaValue(ix, sample) + offset;
      if (v >= ix.bwtLength) v -= ix.bwtLength;
      if (v >= ix.bwtLength) v %= ix.bwtLength;
      positions[t] = v;
    }
  }
}

/* ------------------------------------------------------------------ image build kernels */

/* reference-layout blocks -> device layout; also finds the sentinel's BWT position */
__global__ void relayoutNucKernel(const unsigned long long *__restrict__ ref, unsigned long long numBlocks,
                                  unsigned long long bwtLength, uint4 *__restrict__ out,
                                  unsigned long long *__restrict__ sentinelPos) {
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long blk = t >> 3;
  const unsigned k = (unsigned)t & 7u;
  if (blk >= numBlocks) return;
  const unsigned long long *src = ref + blk * 20ull; /* 160 B = 20 words: planes [3][4], counts [8] */
  const unsigned half = (k & 1u) * 32u;
  const unsigned b0 = (unsigned)(src[0 + (k >> 1)] >> half);
  const unsigned b1 = (unsigned)(src[4 + (k >> 1)] >> half);
  const unsigned b2 = (unsigned)(src[8 + (k >> 1)] >> half);
  const unsigned cw = (unsigned)(src[12 + (k >> 1)] >> half);
  out[blk * 8ull + k] = make_uint4(b0, b1, b2, cw);
  unsigned sentinelBits = b2 & ~b1 & ~b0; /* code 100b, ref src/AwFmLetter.c:44-47 */
  if (sentinelBits) {
    const unsigned long long pos = blk * 256ull + k * 32u + (unsigned)(__ffs((int)sentinelBits) - 1);
    if (pos < bwtLength) *sentinelPos = pos;
  }
}

__global__ void relayoutAminoKernel(const unsigned long long *__restrict__ ref, unsigned long long numBlocks,
                                    unsigned long long bwtLength, uint4 *__restrict__ out,
                                    unsigned long long *__restrict__ sentinelPos) {
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long blk = t >> 3;
  const unsigned k = (unsigned)t & 7u;
  if (blk >= numBlocks) return;
  const unsigned long long *src = ref + blk * 44ull; /* 352 B = 44 words: planes [5][4], counts [24] */
  const unsigned half = (k & 1u) * 32u;
  unsigned b[5];
  for (int j = 0; j < 5; j++) b[j] = (unsigned)(src[4 * j + (k >> 1)] >> half);
  unsigned c[3];
  for (unsigned s = 0; s < 3; s++) {
    const unsigned letter = 3u * k + s;
    c[s] = letter < 21u ? (unsigned)src[20 + letter] : 0u;
  }
  out[blk * 16ull + 2u * k] = make_uint4(b[0], b[1], b[2], b[3]);
  out[blk * 16ull + 2u * k + 1u] = make_uint4(b[4], c[0], c[1], c[2]);
  unsigned sentinelBits = ~(b[0] | b[1] | b[2] | b[3] | b[4]); /* code 00000 */
  while (sentinelBits) {
    const unsigned bit = (unsigned)(__ffs((int)sentinelBits) - 1);
    sentinelBits &= sentinelBits - 1u;
    const unsigned long long pos = blk * 256ull + k * 32u + bit;
    if (pos < bwtLength) *sentinelPos = pos;
  }
}

}  // namespace

/* ------------------------------------------------------------------ host side */

struct AwFmGpuIndex {
  int device = 0;
  bool amino = false;
  DevIndex dev{};
  void *dBlocks = nullptr;
  void *dSeed = nullptr;
  void *dSa = nullptr;
  void *dPrefix = nullptr;
  uint64_t deviceBytes = 0;
  uint64_t numBlocks = 0;
  AwFmGpuKernel kernel = AWFM_GPU_KERNEL_AUTO;
  int numCUs = 256;
  /* grow-only workspace for the host-buffer entry points */
  std::mutex workMutex;
  void *dWork = nullptr;
  size_t workBytes = 0;
  void *hostStage = nullptr; /* pinned staging for small D2H results */
};

namespace {

std::mutex tableMutex;
std::vector<std::pair<const AwFmIndex *, AwFmGpuIndex *>> imageTable;

struct DeviceGuard {
  int previous = -1;
  bool ok = false;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&previous) != hipSuccess) previous = -1;
    ok = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceGuard() {
    if (previous >= 0) (void)hipSetDevice(previous);
  }
};

unsigned gridFor(uint64_t groups, const AwFmGpuIndex *g) {
  const uint64_t blocks = (groups + kGroupsPerBlock - 1) / kGroupsPerBlock;
  const uint64_t cap = (uint64_t)g->numCUs * 8; /* 8 x 256-thread blocks = 32 waves per CU */
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

}  // namespace

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
  if (amino && index->bwtLength >= (1ull << 32)) {
    setError("awfmGpuIndexCreate: amino device layout holds 32-bit base counts; bwtLength must be < 2^32");
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
  const uint64_t numBlocks = awfmNumBlocks(index->bwtLength);
  g->numBlocks = numBlocks;
  const size_t refBytes = numBlocks * awfmBlockBytes(index->config.alphabetType);
  const size_t devBlockBytes = numBlocks * (amino ? 256ull : 128ull);
  const uint64_t seedLen = awfmKmerTableLength(index->config.alphabetType, index->config.kmerLengthInSeedTable);
  const size_t seedBytes = seedLen * sizeof(struct AwFmSearchRange);
  const size_t saBytes = index->suffixArray.compressedByteLength;
  const size_t saAlloc = alignUp(saBytes, 8) + 16;

  auto fail = [&](enum AwFmReturnCode rc) {
    awfmGpuIndexDestroy(g);
    return rc;
  };
  void *dRef = nullptr;
  unsigned long long *dSentinel = nullptr;
#define TRY_OR_FAIL(call, rc)                 \
  do {                                        \
    hipError_t e__ = (call);                  \
    if (e__ != hipSuccess) {                  \
      setError(#call, e__);                   \
      if (dRef) (void)hipFree(dRef);          \
      if (dSentinel) (void)hipFree(dSentinel);\
      return fail(rc);                        \
    }                                         \
  } while (0)

  TRY_OR_FAIL(hipMalloc(&g->dBlocks, devBlockBytes), AwFmAllocationFailure);
  TRY_OR_FAIL(hipMalloc(&g->dSeed, seedBytes ? seedBytes : 16), AwFmAllocationFailure);
  TRY_OR_FAIL(hipMalloc(&g->dSa, saAlloc), AwFmAllocationFailure);
  TRY_OR_FAIL(hipMalloc(&dRef, refBytes), AwFmAllocationFailure);
  TRY_OR_FAIL(hipMalloc((void **)&dSentinel, 8), AwFmAllocationFailure);
  g->deviceBytes = devBlockBytes + seedBytes + saAlloc;

  TRY_OR_FAIL(hipMemcpy(dRef, index->bwtBlockList.asNucleotide, refBytes, hipMemcpyHostToDevice), AwFmGeneralFailure);
  TRY_OR_FAIL(hipMemset(dSentinel, 0, 8), AwFmGeneralFailure);
  {
    const uint64_t threads = numBlocks * 8;
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (amino)
      hipLaunchKernelGGL(relayoutAminoKernel, dim3(grid), dim3(256), 0, 0, (const unsigned long long *)dRef,
                         (unsigned long long)numBlocks, (unsigned long long)index->bwtLength, (uint4 *)g->dBlocks,
                         dSentinel);
    else
      hipLaunchKernelGGL(relayoutNucKernel, dim3(grid), dim3(256), 0, 0, (const unsigned long long *)dRef,
                         (unsigned long long)numBlocks, (unsigned long long)index->bwtLength, (uint4 *)g->dBlocks,
                         dSentinel);
    TRY_OR_FAIL(hipGetLastError(), AwFmGeneralFailure);
  }
  unsigned long long sentinelPos = 0;
  TRY_OR_FAIL(hipMemcpy(&sentinelPos, dSentinel, 8, hipMemcpyDeviceToHost), AwFmGeneralFailure);
  (void)hipFree(dRef);
  dRef = nullptr;
  (void)hipFree(dSentinel);
  dSentinel = nullptr;

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

  DevIndex &d = g->dev;
  d.blocks = (const uint4 *)g->dBlocks;
  d.seed = (const ulonglong2 *)g->dSeed;
  d.sa = (const unsigned long long *)g->dSa;
  d.bwtLength = index->bwtLength;
  d.sentinelPos = sentinelPos;
  d.seedLen = seedLen;
  d.prefixSums = (const unsigned long long *)g->dPrefix;
  d.saRatio = index->config.suffixArrayCompressionRatio;
  d.saShift = 0xFFFFFFFFu;
  if ((d.saRatio & (d.saRatio - 1)) == 0) {
    d.saShift = 0;
    while ((1u << d.saShift) < d.saRatio) d.saShift++;
  }
  d.saWidth = index->suffixArray.valueBitWidth;
  d.seedK = index->config.kmerLengthInSeedTable;
  *out = g;
  return AwFmSuccess;
}

void awfmGpuIndexDestroy(AwFmGpuIndex *g) {
  if (!g) return;
  {
    DeviceGuard guard(g->device);
    if (g->dBlocks) (void)hipFree(g->dBlocks);
    if (g->dSeed) (void)hipFree(g->dSeed);
    if (g->dSa) (void)hipFree(g->dSa);
    if (g->dPrefix) (void)hipFree(g->dPrefix);
    if (g->dWork) (void)hipFree(g->dWork);
    if (g->hostStage) (void)hipHostFree(g->hostStage);
  }
  delete g;
}

AwFmGpuIndex *awfmGpuIndexAcquire(const struct AwFmIndex *index) {
  std::lock_guard<std::mutex> lock(tableMutex);
  for (auto &e : imageTable)
    if (e.first == index) return e.second;
  AwFmGpuIndex *g = nullptr;
  if (awfmGpuIndexCreate(index, -1, &g) != AwFmSuccess) return nullptr;
  imageTable.emplace_back(index, g);
  return g;
}

void awfmGpuIndexRelease(const struct AwFmIndex *index) {
  AwFmGpuIndex *g = nullptr;
  {
    std::lock_guard<std::mutex> lock(tableMutex);
    for (size_t i = 0; i < imageTable.size(); i++)
      if (imageTable[i].first == index) {
        g = imageTable[i].second;
        imageTable.erase(imageTable.begin() + (long)i);
        break;
      }
  }
  awfmGpuIndexDestroy(g);
}

uint64_t awfmGpuIndexDeviceBytes(const AwFmGpuIndex *g) { return g ? g->deviceBytes : 0; }
int awfmGpuIndexDevice(const AwFmGpuIndex *g) { return g ? g->device : -1; }
void awfmGpuIndexSetKernel(AwFmGpuIndex *g, enum AwFmGpuKernel kernel) {
  if (g) g->kernel = kernel;
}

enum AwFmReturnCode awfmGpuSearch(AwFmGpuIndex *g, const uint8_t *dChars, const uint64_t *dOffsets,
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
  const unsigned grid = gridFor(numQueries, g);
  hipStream_t s = (hipStream_t)stream;
  if (g->amino)
    hipLaunchKernelGGL(searchGroup8Kernel<true>, dim3(grid), dim3(kThreads), 0, s, g->dev, dChars,
                       (const unsigned long long *)dOffsets, fixedLength, (unsigned long long)numQueries,
                       (ulonglong2 *)dRanges, dCounts);
  else
    hipLaunchKernelGGL(searchGroup8Kernel<false>, dim3(grid), dim3(kThreads), 0, s, g->dev, dChars,
                       (const unsigned long long *)dOffsets, fixedLength, (unsigned long long)numQueries,
                       (ulonglong2 *)dRanges, dCounts);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  return AwFmSuccess;
}

/* scratch: range lengths (n) + tile sums per level */
uint64_t awfmGpuScanScratchBytes(uint64_t numQueries) {
  uint64_t words = numQueries + 1;
  uint64_t level = numQueries;
  while (level > (uint64_t)kScanTile) {
    level = (level + kScanTile - 1) / kScanTile;
    words += 2 * level + 2; /* sums + their scanned offsets */
  }
  return words * 8 + 64;
}

namespace {
/* exclusive scan of in[0..n) into out[0..n] (out[n] = total), recursive over tiles */
enum AwFmReturnCode scanRecursive(const unsigned long long *in, uint64_t n, unsigned long long *out,
                                  unsigned long long *scratch, hipStream_t s) {
  const uint64_t tiles = (n + kScanTile - 1) / kScanTile;
  if (tiles <= 1) {
    hipLaunchKernelGGL(scanTileKernel, dim3(1), dim3(kScanThreads), 0, s, in, (unsigned long long)n,
                       (const unsigned long long *)nullptr, out, 1);
    AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
    return AwFmSuccess;
  }
  unsigned long long *sums = scratch;
  unsigned long long *offs = scratch + tiles;
  hipLaunchKernelGGL(scanReduceKernel, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, in, (unsigned long long)n, sums);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  const enum AwFmReturnCode rc = scanRecursive(sums, tiles, offs, scratch + 2 * tiles + 2, s);
  if (rc != AwFmSuccess) return rc;
  hipLaunchKernelGGL(scanTileKernel, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, in, (unsigned long long)n,
                     (const unsigned long long *)offs, out, 1);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  return AwFmSuccess;
}
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
  unsigned long long *lengths = (unsigned long long *)dScratch;
  hipLaunchKernelGGL(rangeLengthKernel, dim3((unsigned)((numQueries + 255) / 256)), dim3(256), 0, s,
                     (const ulonglong2 *)dRanges, (unsigned long long)numQueries, lengths);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  const enum AwFmReturnCode rc =
      scanRecursive(lengths, numQueries, (unsigned long long *)dHitOffsets, lengths + numQueries + 1, s);
  if (rc != AwFmSuccess) return rc;
  AWFM_HIP_TRY(hipMemcpyAsync(totalHits, dHitOffsets + numQueries, 8, hipMemcpyDeviceToHost, s), AwFmGeneralFailure);
  AWFM_HIP_TRY(hipStreamSynchronize(s), AwFmGeneralFailure);
  return AwFmSuccess;
}

enum AwFmReturnCode awfmGpuLocate(AwFmGpuIndex *g, const struct AwFmSearchRange *dRanges,
                                  const uint64_t *dHitOffsets, uint64_t numQueries, uint64_t totalHits,
                                  uint64_t *dPositions, void *stream) {
  if (!g) {
    setError("awfmGpuLocate: null image");
    return AwFmNullPtrError;
  }
  if (numQueries == 0 || totalHits == 0) return AwFmSuccess;
  if (!dRanges || !dHitOffsets || !dPositions) {
    setError("awfmGpuLocate: null argument");
    return AwFmNullPtrError;
  }
  DeviceGuard guard(g->device);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(expandHitsKernel, dim3((unsigned)((numQueries + 255) / 256)), dim3(256), 0, s,
                     (const ulonglong2 *)dRanges, (const unsigned long long *)dHitOffsets,
                     (unsigned long long)numQueries, (unsigned long long *)dPositions);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
  const unsigned grid = gridFor(totalHits, g);
  if (g->amino)
    hipLaunchKernelGGL(locateGroup8Kernel<true>, dim3(grid), dim3(kThreads), 0, s, g->dev,
                       (unsigned long long)totalHits, (unsigned long long *)dPositions);
  else
    hipLaunchKernelGGL(locateGroup8Kernel<false>, dim3(grid), dim3(kThreads), 0, s, g->dev,
                       (unsigned long long)totalHits, (unsigned long long *)dPositions);
  AWFM_HIP_TRY(hipGetLastError(), AwFmGeneralFailure);
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
  enum AwFmReturnCode rc = ensureWork(g, l.total);
  if (rc != AwFmSuccess) return rc;
  uint8_t *w = (uint8_t *)g->dWork;
  AWFM_HIP_TRY(hipMemcpy(w + l.chars, chars, totalChars, hipMemcpyHostToDevice), AwFmGeneralFailure);
  if (offsets)
    AWFM_HIP_TRY(hipMemcpy(w + l.offsets, offsets, (numQueries + 1) * 8, hipMemcpyHostToDevice), AwFmGeneralFailure);
  rc = awfmGpuSearch(g, w + l.chars, offsets ? (const uint64_t *)(w + l.offsets) : nullptr, fixedLength, numQueries,
                     (struct AwFmSearchRange *)(w + l.ranges), (uint32_t *)(w + l.counts), nullptr);
  if (rc != AwFmSuccess) return rc;
  AWFM_HIP_TRY(hipDeviceSynchronize(), AwFmGeneralFailure);
  if (ranges) AWFM_HIP_TRY(hipMemcpy(ranges, w + l.ranges, numQueries * 16, hipMemcpyDeviceToHost), AwFmGeneralFailure);
  if (counts) AWFM_HIP_TRY(hipMemcpy(counts, w + l.counts, numQueries * 4, hipMemcpyDeviceToHost), AwFmGeneralFailure);
  return AwFmSuccess;
}

enum AwFmReturnCode awfmGpuLocateHost(AwFmGpuIndex *g, const uint8_t *chars, const uint64_t *offsets,
                                      uint32_t fixedLength, uint64_t numQueries, struct AwFmSearchRange *ranges,
                                      uint64_t *hitOffsets, uint64_t **positions) {
  if (!g || !chars || !hitOffsets || !positions) {
    setError("awfmGpuLocateHost: null argument");
    return AwFmNullPtrError;
  }
  *positions = nullptr;
  hitOffsets[0] = 0;
  if (numQueries == 0) return AwFmSuccess;
  DeviceGuard guard(g->device);
  std::lock_guard<std::mutex> lock(g->workMutex);
  const uint64_t totalChars = offsets ? offsets[numQueries] : numQueries * (uint64_t)fixedLength;
  const HostBatchLayout l = layoutFor(numQueries, totalChars, offsets != nullptr, true);
  enum AwFmReturnCode rc = ensureWork(g, l.total);
  if (rc != AwFmSuccess) return rc;
  uint8_t *w = (uint8_t *)g->dWork;
  AWFM_HIP_TRY(hipMemcpy(w + l.chars, chars, totalChars, hipMemcpyHostToDevice), AwFmGeneralFailure);
  if (offsets)
    AWFM_HIP_TRY(hipMemcpy(w + l.offsets, offsets, (numQueries + 1) * 8, hipMemcpyHostToDevice), AwFmGeneralFailure);
  struct AwFmSearchRange *dRanges = (struct AwFmSearchRange *)(w + l.ranges);
  uint64_t *dHitOffsets = (uint64_t *)(w + l.hitOffsets);
  rc = awfmGpuSearch(g, w + l.chars, offsets ? (const uint64_t *)(w + l.offsets) : nullptr, fixedLength, numQueries,
                     dRanges, nullptr, nullptr);
  if (rc != AwFmSuccess) return rc;
  uint64_t totalHits = 0;
  rc = awfmGpuHitOffsets(g, dRanges, numQueries, dHitOffsets, w + l.scratch, &totalHits, nullptr);
  if (rc != AwFmSuccess) return rc;
  uint64_t *dPositions = nullptr;
  if (totalHits) {
    AWFM_HIP_TRY(hipMalloc((void **)&dPositions, totalHits * 8), AwFmAllocationFailure);
    rc = awfmGpuLocate(g, dRanges, dHitOffsets, numQueries, totalHits, dPositions, nullptr);
    if (rc == AwFmSuccess && hipDeviceSynchronize() != hipSuccess) {
      setError("awfmGpuLocateHost: locate kernels failed", hipGetLastError());
      rc = AwFmGeneralFailure;
    }
    if (rc != AwFmSuccess) {
      (void)hipFree(dPositions);
      return rc;
    }
  }
  uint64_t *hostPositions = (uint64_t *)malloc((totalHits ? totalHits : 1) * 8);
  if (!hostPositions) {
    if (dPositions) (void)hipFree(dPositions);
    setError("awfmGpuLocateHost: host allocation failed");
    return AwFmAllocationFailure;
  }
  hipError_t e = hipSuccess;
  if (totalHits) e = hipMemcpy(hostPositions, dPositions, totalHits * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(hitOffsets, dHitOffsets, (numQueries + 1) * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess && ranges) e = hipMemcpy(ranges, dRanges, numQueries * 16, hipMemcpyDeviceToHost);
  if (dPositions) (void)hipFree(dPositions);
  if (e != hipSuccess) {
    free(hostPositions);
    setError("awfmGpuLocateHost: download failed", e);
    return AwFmGeneralFailure;
  }
  *positions = hostPositions;
  return AwFmSuccess;
}

}  // extern "C"
