/*
 * awfm_gpu_build.hip -- index construction on the GPU.
 *
 * Produces exactly the arrays awFmCreateIndex produces on the host
 * (ref src/AwFmCreate.c:31-137, :281-450; src/AwFmSuffixArray.c:58-112) -- the
 * suffix array of a text is unique, everything after it is arithmetic -- but
 * sized for a 3.1 Gbp text in seconds:
 *   1. sanitise + '$' on the device;
 *   2. suffix array: every suffix gets a 64-bit key of its first 64/b
 *      characters (b bits per dense character code), (key, position) pairs are
 *      radix sorted (rocPRIM onesweep), and suffixes whose keys tie are refined
 *      by prefix doubling (Larsson-Sadakane) on the tied subset only, using the
 *      inverse suffix array as rank;
 *   3. BWT bit planes by wave ballots, per-block letter counts, exclusive scans
 *      -> reference-layout blocks and prefix sums;
 *   4. device-layout blocks (awfm_device.h), seed table level by level with the
 *      same blind backward step the host DFS uses, sampled SA bit-packing;
 *   5. download of the reference-layout arrays into a host AwFmIndex; the
 *      device image is kept and registered for awFmParallelSearch*.
 * Suffix positions and ranks are 32-bit on the device while bwtLength <= 2^32 - 2 and 64-bit beyond (every kernel
 * of the suffix sort is a template over the position type; the doubling keys are then 128-bit, sorted over the bits
 * in use); $AWFM_GPU_DIAG build_wide=1 selects the 64-bit instantiation on any text (tests).
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "awfm_device.h"
#include "awfm_pair.h"

namespace {

constexpr unsigned kSeedGroupsPerBlock = kThreads / 4; /* seedLevelKernel: 4 lanes per table entry */


#define BUILD_TRY(call)                         \
  do {                                          \
    hipError_t err__ = (call);                  \
    if (err__ != hipSuccess) {                  \
      awfmGpuSetHipError(#call, err__);         \
      return false;                             \
    }                                           \
  } while (0)

struct DeviceBuffer {
  void *p = nullptr;
  ~DeviceBuffer() { reset(); }
  bool alloc(size_t bytes) {
    reset();
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e != hipSuccess) {
      awfmGpuSetHipError("hipMalloc (index build)", e);
      p = nullptr;
      return false;
    }
    return true;
  }
  void reset() {
    if (p) (void)hipFree(p);
    p = nullptr;
  }
  void *release() {
    void *r = p;
    p = nullptr;
    return r;
  }
  template <class T>
  T *as() const {
    return (T *)p;
  }
};

typedef unsigned long long u64;
typedef unsigned int u32;

/* grid of a kernel with one thread per element.  A launch holds fewer than 2^32 threads (the dispatch packet counts
 * work-items in 32 bits), so the grid is capped and every such kernel strides over its elements (EACH). */
constexpr u64 kMaxGrid = 1ull << 22;
inline unsigned gridOf(u64 n, unsigned block = 256) {
  const u64 blocks = (n + block - 1) / block;
  return (unsigned)(blocks < kMaxGrid ? (blocks ? blocks : 1) : kMaxGrid);
}
#define EACH(i, n)                                                                                      \
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x, stride__ = (u64)gridDim.x * blockDim.x; i < (n); \
       i += stride__)

/* ---- text ---- */

/* ref src/AwFmCreate.c:452-466 + src/AwFmLetter.c:24-42, :69-79; writes the '$' terminator too */
__global__ void sanitizeKernel(const unsigned char *__restrict__ raw, u64 n, int amino, unsigned char *__restrict__ out) {
  EACH(i, n + 1) {
    if (i == n) {
      out[i] = '$';
      continue;
    }
    const unsigned c = raw[i];
    const unsigned l = c | 0x20u;
    unsigned r;
    if (amino)
      r = (l == 'b' || l == 'x' || c == 0) ? 'z' : c;
    else
      r = (l == 'a' || l == 'c' || l == 'g' || l == 't' || l == 'u' || l == '$') ? l : 'x';
    out[i] = (unsigned char)r;
  }
}

__global__ void byteHistogramKernel(const unsigned char *__restrict__ text, u64 n, u64 *__restrict__ hist) {
  __shared__ unsigned local[256];
  local[threadIdx.x] = 0;
  __syncthreads();
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) atomicAdd(&local[text[i]], 1u);
  __syncthreads();
  if (local[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (u64)local[threadIdx.x]);
}

/* key(i) = dense codes of text[i..i+perKey) packed MSB first, 0 past the end */
constexpr int kKeyTile = 1024;
template <class P>
__global__ void __launch_bounds__(256)
    suffixKeyKernel(const unsigned char *__restrict__ text, u64 n, const unsigned char *__restrict__ codeTable,
                    unsigned bits, unsigned perKey, u64 *__restrict__ keys, P *__restrict__ positions) {
  __shared__ unsigned char sCode[256];
  __shared__ unsigned char sText[kKeyTile + 64];
  sCode[threadIdx.x] = codeTable[threadIdx.x];
  __syncthreads();
  for (u64 base = (u64)blockIdx.x * kKeyTile; base < n; base += (u64)gridDim.x * kKeyTile) {
    for (unsigned t = threadIdx.x; t < kKeyTile + 64; t += 256) {
      const u64 i = base + t;
      sText[t] = i < n ? sCode[text[i]] : 0;
    }
    __syncthreads();
    for (unsigned t = threadIdx.x; t < kKeyTile; t += 256) {
      const u64 i = base + t;
      if (i >= n) break;
      u64 key = 0;
      for (unsigned c = 0; c < perKey; c++) key = (key << bits) | sText[t + c];
      keys[i] = key;
      positions[i] = (P)i;
    }
    __syncthreads();
  }
}

/* ---- suffix array refinement ---- */

__global__ void tiedFlagKernel(const u64 *__restrict__ keys, u64 n, unsigned char *__restrict__ tied) {
  EACH(i, n) {
    const u64 k = keys[i];
    tied[i] = (i > 0 && keys[i - 1] == k) || (i + 1 < n && keys[i + 1] == k);
  }
}

template <class P>
__global__ void inverseSaKernel(const P *__restrict__ sa, u64 n, P *__restrict__ rank) {
  EACH(i, n) rank[sa[i]] = (P)i;
}

/* compact state of the tied suffixes: slot (SA index), val (text position), headSlot candidates */
template <class P>
__global__ void tiedInitKernel(const u64 *__restrict__ keys, const P *__restrict__ sa, const P *__restrict__ slot,
                               u64 t, P *__restrict__ val, P *__restrict__ headSlot) {
  EACH(j, t) {
    const P s = slot[j];
    val[j] = sa[s];
    const bool head = s == 0 || keys[s] != keys[s - 1];
    headSlot[j] = head ? s : (P)0;
  }
}

template <class P>
__global__ void setRankKernel(const P *__restrict__ val, const P *__restrict__ grp, u64 t, P *__restrict__ rank) {
  EACH(j, t) rank[val[j]] = grp[j];
}

/* key of a tied suffix for the round with offset h: (its group, 1 + the rank of the suffix h characters on, 0 past
 * the end), the group in the high half -- 2 x 32 bits for 32-bit positions, 2 x 64 for 64-bit ones */
template <class P>
struct DoublingKey {
  typedef u64 type;
};
template <>
struct DoublingKey<u64> {
  typedef __uint128_t type;
};

template <class P>
__global__ void doublingKeyKernel(const P *__restrict__ val, const P *__restrict__ grp, const P *__restrict__ rank,
                                  u64 t, u64 n, u64 h, typename DoublingKey<P>::type *__restrict__ key2) {
  typedef typename DoublingKey<P>::type K2;
  EACH(j, t) {
    const u64 next = (u64)val[j] + h;
    const u64 second = next < n ? (u64)rank[next] + 1ull : 0ull;
    key2[j] = ((K2)grp[j] << (8u * sizeof(P))) | (K2)second;
  }
}

/* after sorting (key2,val): write the new order into the SA, derive the new group heads */
template <class P>
__global__ void doublingApplyKernel(const typename DoublingKey<P>::type *__restrict__ key2, const P *__restrict__ val,
                                    const P *__restrict__ slot, u64 t, P *__restrict__ sa, P *__restrict__ headSlot) {
  EACH(j, t) {
    sa[slot[j]] = val[j];
    const bool head = j == 0 || key2[j] != key2[j - 1];
    headSlot[j] = head ? slot[j] : (P)0;
  }
}

/* keep[j] = still tied after this round */
template <class K2>
__global__ void stillTiedKernel(const K2 *__restrict__ key2, u64 t, unsigned char *__restrict__ keep) {
  EACH(j, t) {
    const K2 k = key2[j];
    keep[j] = (j > 0 && key2[j - 1] == k) || (j + 1 < t && key2[j + 1] == k);
  }
}

template <class P>
__global__ void compactTriplesKernel(const unsigned char *__restrict__ keep, const P *__restrict__ dest, u64 t,
                                     const P *__restrict__ slotIn, const P *__restrict__ valIn,
                                     const P *__restrict__ grpIn, P *__restrict__ slotOut, P *__restrict__ valOut,
                                     P *__restrict__ grpOut) {
  EACH(j, t) {
    if (!keep[j]) continue;
    const P d = dest[j];
    slotOut[d] = slotIn[j];
    valOut[d] = valIn[j];
    grpOut[d] = grpIn[j];
  }
}

template <class P>
__global__ void flagsToCountsKernel(const unsigned char *__restrict__ flags, u64 t, P *__restrict__ out) {
  EACH(j, t) out[j] = flags[j];
}

template <class P>
struct MaxOp {
  __host__ __device__ P operator()(P a, P b) const { return a > b ? a : b; }
};

/* ---- BWT blocks ---- */

/* one 256-thread workgroup per BWT block: bit planes by wave ballots
 * (ref src/AwFmCreate.c:291-336, :350-395) and the block's letter histogram */
template <bool AMINO, class P>
__global__ void __launch_bounds__(256)
    bwtBlockKernel(const unsigned char *__restrict__ text, const P *__restrict__ sa, u64 n, u64 numBlocks,
                   u64 *__restrict__ refBlocks, u64 *__restrict__ blockCounts, u64 *__restrict__ sentinelPos) {
  constexpr unsigned kPlanes = AMINO ? 5 : 3;
  constexpr unsigned kLetters = AMINO ? 22 : 6;
  constexpr unsigned kWords = AMINO ? 44 : 20; /* u64 words per reference block */
  __shared__ unsigned sCount[4][24];
  for (u64 blk = blockIdx.x; blk < numBlocks; blk += gridDim.x) {
    const u64 i = blk * 256ull + threadIdx.x;
    const bool valid = i < n;
    unsigned letter = 0xFFu, code = 0;
    if (valid) {
      const P p = sa[i];
      if (p == 0) {
        letter = AMINO ? 21u : 5u;
        *sentinelPos = i;
      } else {
        const unsigned c = text[p - 1];
        letter = AMINO ? (c == '$' ? 21u : (unsigned)kAminoTables.letterOfAscii[c & 31u]) : nucLetterIndex(c);
      }
      if (AMINO) {
        /* index -> code, ref src/AwFmLetter.c:81-87 */
        const unsigned char codes[22] = {0x0C, 0x17, 0x03, 0x06, 0x1E, 0x1A, 0x1B, 0x19, 0x15, 0x1C, 0x1D,
                                         0x08, 0x09, 0x04, 0x13, 0x0A, 0x05, 0x16, 0x01, 0x02, 0x1F, 0x00};
        code = codes[letter];
      } else {
        code = (0x421356u >> (4u * letter)) & 7u; /* {6,5,3,1,2,4}, ref src/AwFmLetter.c:44-47 */
      }
    }
    const unsigned wave = threadIdx.x >> 6;
    u64 *dst = refBlocks + blk * kWords;
#pragma unroll
    for (unsigned j = 0; j < kPlanes; j++) {
      const u64 word = __ballot((code >> j) & 1u);
      if ((threadIdx.x & 63u) == 0) dst[4 * j + wave] = word;
    }
    for (unsigned l = 0; l < kLetters; l++) {
      const u64 m = __ballot(letter == l);
      if ((threadIdx.x & 63u) == 0) sCount[wave][l] = (unsigned)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < kLetters)
      blockCounts[(u64)threadIdx.x * numBlocks + blk] =
          (u64)sCount[0][threadIdx.x] + sCount[1][threadIdx.x] + sCount[2][threadIdx.x] + sCount[3][threadIdx.x];
    __syncthreads();
  }
}

/* base occurrences = exclusive scan of the per-block counts, copied into the block headers
 * (ref src/AwFmCreate.c:304-313, :361-370); the two pad counters are zero */
template <bool AMINO>
__global__ void baseOccurrenceKernel(const u64 *__restrict__ scanned, u64 numBlocks, u64 *__restrict__ refBlocks) {
  constexpr unsigned kLetters = AMINO ? 22 : 6;
  constexpr unsigned kCounters = AMINO ? 24 : 8;
  constexpr unsigned kWords = AMINO ? 44 : 20;
  constexpr unsigned kFirst = AMINO ? 20 : 12;
  EACH(t, numBlocks * kCounters) {
    const u64 blk = t / kCounters;
    const unsigned c = (unsigned)(t % kCounters);
    refBlocks[blk * kWords + kFirst + c] = c < kLetters ? scanned[(u64)c * numBlocks + blk] : 0ull;
  }
}

/* ---- seed table ---- */

/* level L+1 from level L: entry e = a * |A|^L + parent, range = blind backward step of the
 * parent's range with letter a -- no validity check (ref src/AwFmCreate.c:419-450) */
template <bool AMINO, bool STOP_AT_INVALID = false>
__global__ void __launch_bounds__(kThreads)
    seedLevelKernel(const DevIndex ix, const ulonglong2 *__restrict__ parentLevel, u64 parentLen, u64 outLen,
                    ulonglong2 *__restrict__ out) {
  constexpr int G = 4; /* one slice of a block per lane */
  __shared__ u64 sC[24];
  __shared__ AminoShared sAmino;
  __shared__ unsigned sMask[(kBlockMask + 1) * kSlices];
  __shared__ u64 sSuper[AMINO ? 1 : kMaxNucSuper * 4];
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  if (AMINO) aminoStageTables(sAmino);
  stageMaskTable(sMask);
  if (!AMINO) nucStageSuper<false>(ix, sSuper);
  __syncthreads();
  const unsigned g = threadIdx.x % G;
  const u64 numGroups = (u64)gridDim.x * kSeedGroupsPerBlock;
  for (u64 e = ((u64)blockIdx.x * kThreads + threadIdx.x) / G; e < outLen; e += numGroups) {
    const unsigned a = (unsigned)(e / parentLen);
    const ulonglong2 r = parentLevel[e % parentLen];
    u64 sp = r.x, ep = r.y;
    /* STOP_AT_INVALID (deeper device table): a query stops at its first invalid range and keeps it
     * (ref src/AwFmParallelSearch.c:293-294), so an invalid parent is inherited unchanged */
    if (!STOP_AT_INVALID || sp <= ep) {
      if (AMINO)
        aminoStepAny<G, false>(ix, sC, sAmino, sMask, g, a, sp, ep);
      else
        nucStepAny<G, false>(ix, sC, sSuper, g, a, sp, ep);
    }
    if (g == 0) out[e] = make_ulonglong2(sp, ep);
  }
}

/* The same level step for the deeper device-only table of a nucleotide image, where a level has 10^8 entries and more:
 * a group of 4 lanes takes U consecutive entries per round and requests the blocks of all of them before it ranks any,
 * so that a round costs one memory latency for U entries (the one-entry-per-round kernel above is bound by that latency:
 * 3.6 s for the two levels from k = 12 to 14 of a 3.1 Gbp index).  Consecutive entries have consecutive parents, whose
 * ranges are neighbours in the BWT: the U steps of a round mostly read the same few lines. */
/* OUT: 0 -- 16-byte entries {sp, ep}; the level is the table itself: 1 -- 8-byte entries {sp, length} (an image below 2^32
 * positions), 2 -- sp36 | length12 << 36 | all sixteen next-step bits, the lengths of 4095 and more in big[sp >> 11]
 * (DevIndex::deepNarrow, awfm_device.h) */
template <int OUT>
__device__ __forceinline__ void deepSeedStore(ulonglong2 *__restrict__ out, u64 at, u64 sp, u64 ep, u64 *__restrict__ big) {
  if (OUT == 1) {
    ((uint2 *)out)[at] = make_uint2((unsigned)sp, (unsigned)(ep + 1ull - sp));
  } else if (OUT == 2) {
    const u64 length = ep + 1ull - sp;
    ((uint2 *)out)[at] = deepWidePack(sp, length, 0xFFFFu);
    if (length >= kDeepWideLengthMask) big[sp >> kDeepWideBigShift] = length;
  } else {
    out[at] = make_ulonglong2(sp, ep);
  }
}
template <int U, int OUT>
__global__ void __launch_bounds__(kThreads)
    deepSeedLevelKernel(const DevIndex ix, const ulonglong2 *__restrict__ parentLevel, u64 parentLen, u64 outLen,
                        ulonglong2 *__restrict__ out, u64 *__restrict__ big) {
  constexpr int G = 4;
  __shared__ u64 sC[24];
  __shared__ u64 sSuper[kMaxNucSuper * 4];
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  nucStageSuper<false>(ix, sSuper);
  __syncthreads();
  const unsigned g = threadIdx.x % G;
  const u64 numGroups = (u64)gridDim.x * kSeedGroupsPerBlock;
  for (u64 base = (((u64)blockIdx.x * kThreads + threadIdx.x) / G) * U; base < outLen; base += numGroups * U) {
    u64 sp[U], ep[U];
    unsigned letter[U];
    bool step[U];
    Piece p0[U], p1[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const u64 e = base + u < outLen ? base + u : outLen - 1; /* the tail repeats the last entry */
      letter[u] = (unsigned)(e / parentLen);
      const ulonglong2 r = parentLevel[e % parentLen];
      sp[u] = r.x;
      ep[u] = r.y;
      /* a query stops at its first invalid range and keeps it (ref src/AwFmParallelSearch.c:293-294) */
      step[u] = sp[u] <= ep[u];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const u64 blk0 = step[u] ? (sp[u] - 1ull) >> kBlockShift : 0ull, blk1 = step[u] ? ep[u] >> kBlockShift : 0ull;
      p0[u] = *(const Piece *)(ix.blocks + (blk0 * kSlices + g));
      p1[u] = *(const Piece *)(ix.blocks + (blk1 * kSlices + g));
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      u64 a = sp[u], b = ep[u];
      if (step[u]) nucStepAnyRank<G, false>(ix, sC, sSuper, g, letter[u], &p0[u], &p1[u], a, b);
      if (g == 0 && base + u < outLen) deepSeedStore<OUT>(out, base + u, a, b, big);
    }
  }
}

/* TWO levels in one: entry code * parentLen + p of level L + 2 is the level-L entry p after the pair step `code` = 4 c0 + c1
 * (c1 prepended first, then c0: the index puts the leftmost character first) through the pair image (awfm_pair.h), EXACT --
 * a range that dies inside the pair ends in the empty range the letter-by-letter stepping ends in, as in the general search
 * kernel, whose three lines these are.  The level in between is never written: a depth-16 table is built 12 -> 14 -> 16 and
 * the 17 GB of level 15 -- a quarter of the construction's allocations, which are most of its time -- are not needed.
 * Images that have their pair image; NARROW: 32-bit positions (awfmImageNarrow). */
template <int OUT, bool NARROW>
__global__ void __launch_bounds__(kThreads)
    deepSeedPairLevelKernel(const DevIndex ix, const ulonglong2 *__restrict__ parentLevel, u64 parentLen, u64 outLen,
                            ulonglong2 *__restrict__ out, u64 *__restrict__ big) {
  constexpr int G = 4;
  typedef typename PositionType<NARROW>::type pos_t;
  __shared__ u64 sC[24];
  __shared__ unsigned sMask[(kBlockMask + 1) * kSlices];
  __shared__ u64 sSuper[NARROW ? 1 : kMaxNucSuper * 4];
  __shared__ u64 sPairC[16];
  extern __shared__ unsigned sPairSuper[];
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  stageMaskTable(sMask);
  nucStageSuper<NARROW>(ix, sSuper);
  pairStageTables<NARROW, 16u>(ix, sPairC, sPairSuper);
  __syncthreads();
  const unsigned gl = threadIdx.x % G;
  const u64 numGroups = (u64)gridDim.x * kSeedGroupsPerBlock;
  for (u64 e = ((u64)blockIdx.x * kThreads + threadIdx.x) / G; e < outLen; e += numGroups) {
    const unsigned code = (unsigned)(e / parentLen);
    const ulonglong2 r = parentLevel[e % parentLen];
    pos_t sp = (pos_t)r.x, ep = (pos_t)r.y;
    if (r.x <= r.y) { /* a query stops at its first invalid range and keeps it (ref src/AwFmParallelSearch.c:293-294) */
      const unsigned c2 = code & 3u, c1 = code >> 2;
      const PairStep did = pairSearchStep<NARROW, true>(ix, sPairC, sPairSuper, sMask, gl, code, sp, ep, sC);
      if (did == kPairFlagged) nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, gl, c2, sp, ep);
      if (did != kPairStepped && sp <= ep) nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, gl, c1, sp, ep);
    }
    if (gl == 0) {
      /* (an empty range is {sp, sp - 1}, sp >= 1; a parent that was empty already is passed on as it is) */
      if (OUT == 0 && r.x > r.y) out[e] = r;
      else deepSeedStore<OUT>(out, e, r.x <= r.y ? (u64)sp : r.x, r.x <= r.y ? (u64)ep : r.y, big);
    }
  }
}

/* ---- one table per k-mer length below the deeper table's (awfmGpuBuildLengthTables) ----
 * Level d holds, for every string X of d letters (index: leftmost letter first, as in the seed table), the 8-byte entry
 * {sp, length} of its range -- what the reference reaches for a k-mer of exactly d characters: from the letter range of the
 * last character and d - 1 steps when d is below the seed table's length (ref src/AwFmSearch.c:485-520), the seed table's
 * entry at d = seedK (ref src/AwFmKmerTable.c:4-51), that entry and d - seedK steps that stop at the first empty range above
 * it (ref src/AwFmParallelSearch.c:273-313).  A length of 0 says "no hit" (all a hits-only search reports of it).  Level
 * d + 1 from level d: entry letter * 4^d + p is entry p after one backward step with `letter`; one group of 4 lanes per entry
 * (nucFastStep: the step of the search kernels), consecutive entries have consecutive parents, whose ranges are neighbours
 * in the BWT.  Images below 2^32 positions. */
/* WIDE (round 6): the image runs 64-bit positions and the entries are sp36 | length28 << 36, the lengths of 2^28 - 1 and more
 * in big[(d - 1) * 512 + (sp >> 27)] (awfm_device.h: lengthEntryOpen); `level`: the d of the entries written */
template <bool WIDE>
__device__ __forceinline__ void lengthStore(uint2 *__restrict__ out, u64 at, u64 sp, u64 length, u64 *__restrict__ big, unsigned level) {
  if (WIDE) {
    out[at] = lengthWidePack(sp, length);
    if (length >= kLengthWideMask) big[(level - 1u) * kLengthWideBigStride + (unsigned)(sp >> kLengthWideBigShift)] = length;
  } else {
    out[at] = make_uint2((unsigned)sp, (unsigned)length);
  }
}
template <bool WIDE>
__global__ void __launch_bounds__(kThreads)
    lengthLevelKernel(const DevIndex ix, const uint2 *__restrict__ parentLevel, u64 parentLen, u64 outLen, uint2 *__restrict__ out,
                      u64 *__restrict__ big, unsigned level) {
  constexpr int G = 4;
  constexpr bool NARROW = !WIDE;
  typedef typename PositionType<NARROW>::type pos_t;
  __shared__ u64 sC[24];
  __shared__ unsigned sMask[(kBlockMask + 1) * kSlices];
  __shared__ u64 sSuper[NARROW ? 1 : kMaxNucSuper * 4];
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  stageMaskTable(sMask);
  nucStageSuper<NARROW>(ix, sSuper);
  __syncthreads();
  const unsigned gl = threadIdx.x % G;
  const u64 numGroups = (u64)gridDim.x * kSeedGroupsPerBlock;
  DevIndex view = ix; /* (the parent level is read in the format this construction writes) */
  view.deepNarrow = WIDE ? 2u : 1u;
  view.lengthBig = big;
  for (u64 e = ((u64)blockIdx.x * kThreads + threadIdx.x) / G; e < outLen; e += numGroups) {
    const unsigned letter = (unsigned)(e / parentLen);
    const ulonglong2 r = lengthEntryOpen(view, level - 1u, parentLevel[e % parentLen]);
    pos_t sp = (pos_t)r.x, ep = (pos_t)(r.x + r.y - 1ull);
    if (r.y != 0ull) nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, gl, letter, sp, ep);
    if (gl == 0) lengthStore<WIDE>(out, e, (u64)sp, r.y != 0ull && sp <= ep ? (u64)(ep + (pos_t)1 - sp) : 0ull, big, level);
  }
}

/* level 1: the letter ranges; level seedK: the index's own table in 8-byte form */
template <bool WIDE>
__global__ void lengthLettersKernel(const DevIndex ix, uint2 *__restrict__ out, u64 *__restrict__ big) {
  if (threadIdx.x < 4u) {
    const u64 first = ix.prefixSums[threadIdx.x], next = ix.prefixSums[threadIdx.x + 1u];
    lengthStore<WIDE>(out, threadIdx.x, first, next - first, big, 1u);
  }
}
template <bool WIDE>
__global__ void __launch_bounds__(256) lengthFromSeedKernel(const ulonglong2 *__restrict__ seed, u64 len, uint2 *__restrict__ out,
                                                            u64 *__restrict__ big, unsigned level) {
  for (u64 e = (u64)blockIdx.x * 256u + threadIdx.x; e < len; e += (u64)gridDim.x * 256u) {
    const ulonglong2 r = seed[e];
    lengthStore<WIDE>(out, e, r.x, r.x <= r.y ? r.y + 1ull - r.x : 0ull, big, level);
  }
}

/* The same level step for the amino alphabet: entry letter * parentLen + p of the level is the level-below entry p after
 * one backward step with `letter` (0..19: the index puts the leftmost character first, ref src/AwFmKmerTable.c:37-51), or
 * that entry unchanged when its range is already empty.  One group of 4 lanes per entry (aminoStepAny: the step of the
 * search kernel); consecutive entries have consecutive parents, whose ranges are neighbours in the BWT. */
template <int OUT> /* 0: {sp, ep}; 1: {sp32, length32}; 2: aminoWidePack with every next-letter bit set, the long lengths in big[sp >> 7] */
__global__ void __launch_bounds__(kThreads)
    aminoDeepSeedLevelKernel(const DevIndex ix, const ulonglong2 *__restrict__ parentLevel, u64 parentLen, u64 outLen,
                             ulonglong2 *__restrict__ out, u64 *__restrict__ big) {
  constexpr int G = 4;
  __shared__ u64 sC[24];
  __shared__ AminoShared sAmino;
  __shared__ unsigned sMask[(kBlockMask + 1) * kSlices];
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  aminoStageTables(sAmino);
  stageMaskTable(sMask);
  __syncthreads();
  const unsigned g = threadIdx.x % G;
  const u64 numGroups = (u64)gridDim.x * (kThreads / G);
  for (u64 e = ((u64)blockIdx.x * kThreads + threadIdx.x) / G; e < outLen; e += numGroups) {
    const unsigned letter = (unsigned)(e / parentLen);
    const ulonglong2 r = parentLevel[e % parentLen];
    u64 sp = r.x, ep = r.y;
    /* a query stops at its first invalid range and keeps it (ref src/AwFmParallelSearch.c:293-294) */
    if (sp <= ep) aminoStepAny<G, false>(ix, sC, sAmino, sMask, g, letter, sp, ep);
    if (g == 0) {
      if (OUT == 1) {
        ((uint2 *)out)[e] = make_uint2((unsigned)sp, (unsigned)(ep + 1ull - sp));
      } else if (OUT == 2) {
        const u64 length = ep + 1ull - sp;
        ((uint2 *)out)[e] = aminoWidePack(sp, length, 0xFFFFFu);
        if (length >= kAminoWideLengthMask) big[sp >> kAminoWideBigShift] = length;
      } else {
        out[e] = make_ulonglong2(sp, ep);
      }
    }
  }
}

/* ---- sampled SA ---- */

/* 64-bit word j of the little-endian bit stream of samples SA[s*ratio], `width` bits each
 * (ref src/AwFmSuffixArray.c:58-112) */
template <class P>
__global__ void packSampledSaKernel(const P *__restrict__ sa, u64 samples, unsigned ratio, unsigned width,
                                    u64 words, u64 *__restrict__ out) {
  EACH(j, words) {
    const u64 firstBit = j * 64ull;
    u64 word = 0;
    for (u64 s = firstBit / width; s < samples && s * width < firstBit + 64ull; s++) {
      const u64 v = sa[s * ratio];
      const long long shift = (long long)(s * width) - (long long)firstBit;
      word |= shift >= 0 ? v << shift : v >> (-shift);
    }
    out[j] = word;
  }
}

/* ---- rocPRIM wrappers ---- */

/* sorted (keysIn, valsIn) into (keysOut, valsOut); BOTH pairs of buffers are working space (rocPRIM's double-buffer form: no
 * third copy of the keys and values in the temporary storage, which at 64-bit positions was 16 of the construction's 49
 * bytes of HBM per text position -- a two-strand human genome did not fit one GPU with it) */
template <class K, class V>
bool sortPairs(K *keysIn, K *keysOut, V *valsIn, V *valsOut, u64 n, unsigned endBit, DeviceBuffer &temp,
               size_t &tempBytes) {
  rocprim::double_buffer<K> keys(keysIn, keysOut);
  rocprim::double_buffer<V> vals(valsIn, valsOut);
  size_t need = 0;
  BUILD_TRY(rocprim::radix_sort_pairs(nullptr, need, keys, vals, (size_t)n, 0u, endBit, (hipStream_t)0));
  if (need > tempBytes) {
    if (!temp.alloc(need)) return false;
    tempBytes = need;
  }
  BUILD_TRY(rocprim::radix_sort_pairs(temp.p, need, keys, vals, (size_t)n, 0u, endBit, (hipStream_t)0));
  if (keys.current() != keysOut) BUILD_TRY(hipMemcpyAsync(keysOut, keys.current(), n * sizeof(K), hipMemcpyDeviceToDevice, (hipStream_t)0));
  if (vals.current() != valsOut) BUILD_TRY(hipMemcpyAsync(valsOut, vals.current(), n * sizeof(V), hipMemcpyDeviceToDevice, (hipStream_t)0));
  return true;
}

template <class P>
bool maxScan(const P *in, P *out, u64 n, DeviceBuffer &temp, size_t &tempBytes) {
  size_t need = 0;
  BUILD_TRY(rocprim::inclusive_scan(nullptr, need, in, out, (size_t)n, MaxOp<P>(), (hipStream_t)0));
  if (need > tempBytes) {
    if (!temp.alloc(need)) return false;
    tempBytes = need;
  }
  BUILD_TRY(rocprim::inclusive_scan(temp.p, need, in, out, (size_t)n, MaxOp<P>(), (hipStream_t)0));
  return true;
}

template <class T>
bool exclusiveSum(const T *in, T *out, u64 n, DeviceBuffer &temp, size_t &tempBytes) {
  size_t need = 0;
  BUILD_TRY(rocprim::exclusive_scan(nullptr, need, in, out, (T)0, (size_t)n, rocprim::plus<T>(), (hipStream_t)0));
  if (need > tempBytes) {
    if (!temp.alloc(need)) return false;
    tempBytes = need;
  }
  BUILD_TRY(rocprim::exclusive_scan(temp.p, need, in, out, (T)0, (size_t)n, rocprim::plus<T>(), (hipStream_t)0));
  return true;
}

/* indices i in [0,n) with flags[i] != 0 -> out, count -> *countHost */
template <class P>
bool selectFlagged(const unsigned char *flags, u64 n, P *out, u64 *countHost, DeviceBuffer &temp, size_t &tempBytes) {
  DeviceBuffer dCount;
  if (!dCount.alloc(sizeof(u64))) return false;
  rocprim::counting_iterator<P> ids((P)0);
  size_t need = 0;
  BUILD_TRY(rocprim::select(nullptr, need, ids, flags, out, dCount.as<u64>(), (size_t)n, (hipStream_t)0));
  if (need > tempBytes) {
    if (!temp.alloc(need)) return false;
    tempBytes = need;
  }
  BUILD_TRY(rocprim::select(temp.p, need, ids, flags, out, dCount.as<u64>(), (size_t)n, (hipStream_t)0));
  BUILD_TRY(hipMemcpy(countHost, dCount.p, sizeof(u64), hipMemcpyDeviceToHost));
  return true;
}

/* suffix array of dText[0..n) into dSa (positions of type P); dText ends with the unique '$' */
template <class P>
bool buildSuffixArray(const unsigned char *dText, u64 n, P *dSa, bool verbose) {
  typedef typename DoublingKey<P>::type K2;
  constexpr size_t W = sizeof(P);
  /* dense codes from the byte histogram */
  DeviceBuffer dHist, dCode;
  if (!dHist.alloc(256 * sizeof(u64)) || !dCode.alloc(256)) return false;
  BUILD_TRY(hipMemset(dHist.p, 0, 256 * sizeof(u64)));
  hipLaunchKernelGGL(byteHistogramKernel, dim3(2048), dim3(256), 0, 0, dText, n, dHist.as<u64>());
  u64 hist[256];
  BUILD_TRY(hipMemcpy(hist, dHist.p, sizeof hist, hipMemcpyDeviceToHost));
  unsigned char code[256];
  unsigned distinct = 0;
  for (int c = 0; c < 256; c++) code[c] = hist[c] ? (unsigned char)++distinct : 0;
  unsigned bits = 1;
  while ((1u << bits) <= distinct) bits++;
  const unsigned perKey = 64 / bits;
  BUILD_TRY(hipMemcpy(dCode.p, code, 256, hipMemcpyHostToDevice));
  /* bits of a rank + 1 (<= n): what the low half of a doubling key holds */
  unsigned rankBits = 1;
  while (rankBits < 64 && (n >> rankBits) != 0) rankBits++;
  const unsigned doublingEndBit = (unsigned)(8 * W) + (W == 4 ? 32u : rankBits);

  DeviceBuffer temp;
  size_t tempBytes = 0;
  DeviceBuffer dSlot;
  u64 tied = 0;
  DeviceBuffer dRank;
  {
    DeviceBuffer keysA, keysB, valsB;
    if (!keysA.alloc(n * 8) || !keysB.alloc(n * 8) || !valsB.alloc(n * W)) return false;
    hipLaunchKernelGGL(suffixKeyKernel<P>, dim3(gridOf(n, kKeyTile)), dim3(256), 0, 0, dText, n,
                       dCode.as<unsigned char>(), bits, perKey, keysA.as<u64>(), valsB.as<P>());
    BUILD_TRY(hipGetLastError());
    if (!sortPairs(keysA.as<u64>(), keysB.as<u64>(), valsB.as<P>(), dSa, n, bits * perKey, temp, tempBytes)) return false;
    keysA.reset();
    valsB.reset();
    /* which suffixes still tie on their first perKey characters */
    DeviceBuffer dTied;
    if (!dTied.alloc(n)) return false;
    hipLaunchKernelGGL(tiedFlagKernel, dim3(gridOf(n)), dim3(256), 0, 0, keysB.as<u64>(), n, dTied.as<unsigned char>());
    BUILD_TRY(hipGetLastError());
    if (!dSlot.alloc(n * W)) return false; /* worst case: everything ties */
    if (!selectFlagged(dTied.as<unsigned char>(), n, dSlot.as<P>(), &tied, temp, tempBytes)) return false;
    if (verbose)
      fprintf(stderr, "[awfm build] %llu suffixes (%zu-bit positions), %u bits/char, %u chars/key, %llu tied after the key sort\n",
              n, 8 * W, bits, perKey, tied);
    if (tied == 0) return true;
    if (!dRank.alloc(n * W)) return false;
    hipLaunchKernelGGL(inverseSaKernel<P>, dim3(gridOf(n)), dim3(256), 0, 0, dSa, n, dRank.as<P>());
    BUILD_TRY(hipGetLastError());
    /* compact tied state, group heads from the sorted keys */
    DeviceBuffer dVal, dHead, dGrp;
    if (!dVal.alloc(tied * W) || !dHead.alloc(tied * W) || !dGrp.alloc(tied * W)) return false;
    hipLaunchKernelGGL(tiedInitKernel<P>, dim3(gridOf(tied)), dim3(256), 0, 0, keysB.as<u64>(), dSa, dSlot.as<P>(), tied,
                       dVal.as<P>(), dHead.as<P>());
    BUILD_TRY(hipGetLastError());
    keysB.reset();
    dTied.reset();
    if (!maxScan(dHead.as<P>(), dGrp.as<P>(), tied, temp, tempBytes)) return false;
    hipLaunchKernelGGL(setRankKernel<P>, dim3(gridOf(tied)), dim3(256), 0, 0, dVal.as<P>(), dGrp.as<P>(), tied,
                       dRank.as<P>());
    BUILD_TRY(hipGetLastError());

    /* prefix doubling on the tied subset */
    DeviceBuffer key2A, key2B, valOut, keep, dest, slot2, val2, grp2;
    u64 t = tied;
    DeviceBuffer curSlot, curVal, curGrp;
    curSlot.p = dSlot.release();
    curVal.p = dVal.release();
    curGrp.p = dGrp.release();
    for (u64 h = perKey; t > 0; h *= 2) {
      if (h >= 2 * n + perKey) { /* every suffix is unique by its first n characters: cannot happen on sound state */
        awfmGpuSetError("awfmGpuCreateIndex: the suffix sort did not converge");
        return false;
      }
      if (!key2A.alloc(t * sizeof(K2)) || !key2B.alloc(t * sizeof(K2)) || !valOut.alloc(t * W) || !keep.alloc(t) ||
          !dest.alloc(t * W))
        return false;
      hipLaunchKernelGGL(doublingKeyKernel<P>, dim3(gridOf(t)), dim3(256), 0, 0, curVal.as<P>(), curGrp.as<P>(),
                         dRank.as<P>(), t, n, h, key2A.as<K2>());
      BUILD_TRY(hipGetLastError());
      if (!sortPairs(key2A.as<K2>(), key2B.as<K2>(), curVal.as<P>(), valOut.as<P>(), t, doublingEndBit, temp, tempBytes))
        return false;
      hipLaunchKernelGGL(doublingApplyKernel<P>, dim3(gridOf(t)), dim3(256), 0, 0, key2B.as<K2>(), valOut.as<P>(),
                         curSlot.as<P>(), t, dSa, dHead.as<P>());
      BUILD_TRY(hipGetLastError());
      if (!maxScan(dHead.as<P>(), curGrp.as<P>(), t, temp, tempBytes)) return false;
      hipLaunchKernelGGL(setRankKernel<P>, dim3(gridOf(t)), dim3(256), 0, 0, valOut.as<P>(), curGrp.as<P>(), t,
                         dRank.as<P>());
      hipLaunchKernelGGL(stillTiedKernel<K2>, dim3(gridOf(t)), dim3(256), 0, 0, key2B.as<K2>(), t,
                         keep.as<unsigned char>());
      BUILD_TRY(hipGetLastError());
      /* compaction: exclusive sum of keep flags */
      DeviceBuffer keepCounts;
      if (!keepCounts.alloc(t * W)) return false;
      hipLaunchKernelGGL(flagsToCountsKernel<P>, dim3(gridOf(t)), dim3(256), 0, 0, keep.as<unsigned char>(), t,
                         keepCounts.as<P>());
      if (!exclusiveSum<P>(keepCounts.as<P>(), dest.as<P>(), t, temp, tempBytes)) return false;
      P lastDest = 0;
      unsigned char lastKeep = 0;
      BUILD_TRY(hipMemcpy(&lastDest, dest.as<P>() + (t - 1), W, hipMemcpyDeviceToHost));
      BUILD_TRY(hipMemcpy(&lastKeep, keep.as<unsigned char>() + (t - 1), 1, hipMemcpyDeviceToHost));
      const u64 newT = (u64)lastDest + lastKeep;
      if (verbose) fprintf(stderr, "[awfm build]   doubling h=%llu: %llu tied -> %llu\n", h, t, newT);
      if (newT == 0) break;
      if (!slot2.alloc(newT * W) || !val2.alloc(newT * W) || !grp2.alloc(newT * W)) return false;
      hipLaunchKernelGGL(compactTriplesKernel<P>, dim3(gridOf(t)), dim3(256), 0, 0, keep.as<unsigned char>(), dest.as<P>(),
                         t, curSlot.as<P>(), valOut.as<P>(), curGrp.as<P>(), slot2.as<P>(), val2.as<P>(), grp2.as<P>());
      BUILD_TRY(hipGetLastError());
      BUILD_TRY(hipDeviceSynchronize());
      curSlot.reset();
      curVal.reset();
      curGrp.reset();
      curSlot.p = slot2.release();
      curVal.p = val2.release();
      curGrp.p = grp2.release();
      if (!dHead.alloc(newT * W)) return false;
      t = newT;
    }
  }
  BUILD_TRY(hipDeviceSynchronize());
  return true;
}

/* what the rest of the build reads the suffix array for: the BWT blocks and the sampled, bit-packed SA */
template <class P>
bool launchBwtBlocks(bool amino, const unsigned char *dText, const void *dSa, u64 n, u64 numBlocks, u64 *refBlocks,
                     u64 *blockCounts, u64 *sentinelPos) {
  if (amino)
    hipLaunchKernelGGL((bwtBlockKernel<true, P>), dim3(gridOf(numBlocks, 1)), dim3(256), 0, 0, dText, (const P *)dSa, n,
                       numBlocks, refBlocks, blockCounts, sentinelPos);
  else
    hipLaunchKernelGGL((bwtBlockKernel<false, P>), dim3(gridOf(numBlocks, 1)), dim3(256), 0, 0, dText, (const P *)dSa, n,
                       numBlocks, refBlocks, blockCounts, sentinelPos);
  BUILD_TRY(hipGetLastError());
  return true;
}

template <class P>
bool launchPackSampledSa(const void *dSa, u64 samples, unsigned ratio, unsigned width, u64 words, u64 *out) {
  hipLaunchKernelGGL(packSampledSaKernel<P>, dim3(gridOf(words)), dim3(256), 0, 0, (const P *)dSa, samples, ratio, width,
                     words, out);
  BUILD_TRY(hipGetLastError());
  return true;
}

}  // namespace

/* Deeper seed table for the device image: level seedK is the index's own table; level L+1 extends
 * every level-L entry by one more (prepended) letter with the search path's stop-at-first-invalid rule. */
bool awfmGpuBuildDeepSeedTable(const AwFmGpuIndex *g, unsigned deepK, void **tableOut, uint64_t *bytesOut, uint64_t *peakBytesOut,
                               double *allocSecondsOut, unsigned *formatOut, void **bigOut) {
  *tableOut = nullptr;
  *bytesOut = 0;
  if (peakBytesOut) *peakBytesOut = 0;
  if (allocSecondsOut) *allocSecondsOut = 0.0;
  if (formatOut) *formatOut = 0;
  if (bigOut) *bigOut = nullptr;
  u64 curBytes = 0; /* the level the next one is made from (0: the index's own table) */
  const unsigned K = g->dev.seedK;
  /* nucleotide: up to 16 characters (2^32 entries); amino: up to 7 (20^7 = 1.28 * 10^9 entries, the index a 32-bit sum) */
  if (deepK <= K || deepK > (g->amino ? 7u : 16u) || K == 0) {
    awfmGpuSetError("deep seed table: seedK < deepK <= 16 (nucleotide) / 7 (amino)");
    return false;
  }
  const unsigned card = g->amino ? 20u : 4u;
  DeviceGuard guard(g->device);
  u64 len = 1;
  for (unsigned i = 0; i < K; i++) len *= card;
  DeviceBuffer cur, nxt, big;
  const ulonglong2 *parent = g->dev.seed;
  const bool narrow = awfmImageNarrow(g);
  /* the table's entries (DevIndex::deepNarrow): 8 bytes {sp, length} where the image runs 32-bit positions (every amino
   * image below 2^32), 8 bytes sp36 | length12 | next16 (nucleotide) or sp36 | length8 | next20 (amino) for the images that run
   * 64-bit ones, up to 2^36 positions (round 6), 16 bytes {sp, ep} for everything else */
  const unsigned format = g->dev.bwtLength < (1ull << 32) && narrow ? 1u
                          : (formatOut && bigOut && g->dev.bwtLength < (1ull << kDeepWideMaxBits) ? 2u : 0u);
  if (format == 2u) {
    const size_t bigBytes = ((size_t)(g->dev.bwtLength >> (g->amino ? kAminoWideBigShift : kDeepWideBigShift)) + 2u) * 8u;
    if (!big.alloc(bigBytes)) return false;
    BUILD_TRY(awfmGpuSetupMemset(big.p, 0, bigBytes));
  }
  /* two levels per pass through the pair image where the image has one (deepSeedPairLevelKernel) */
  const bool pairLevels = !g->amino && g->dev.pairBlocks;
  const bool pairSuperInLds = pairLevels && narrow && awfmPairSuperInLds(g);
  for (unsigned L = K; L < deepK;) {
    const unsigned levels = pairLevels && deepK - L >= 2u ? 2u : 1u;
    u64 outLen = len;
    for (unsigned i = 0; i < levels; i++) outLen *= card;
    const unsigned out = L + levels == deepK ? format : 0u; /* the table itself */
    const u64 entryBytes = out ? 8u : 16u;
    struct timespec ta, tb, tc;
    clock_gettime(CLOCK_MONOTONIC, &ta);
    if (!nxt.alloc(outLen * entryBytes)) return false;
    clock_gettime(CLOCK_MONOTONIC, &tb);
    if (allocSecondsOut) *allocSecondsOut += (double)(tb.tv_sec - ta.tv_sec) + 1e-9 * (double)(tb.tv_nsec - ta.tv_nsec);
    if (peakBytesOut && curBytes + outLen * entryBytes > *peakBytesOut) *peakBytesOut = curBytes + outLen * entryBytes;
    constexpr int kUnroll = 4;
    const u64 blocks = (outLen + kSeedGroupsPerBlock * kUnroll - 1) / (kSeedGroupsPerBlock * kUnroll);
    const u64 resident = (u64)g->numCUs * 8u; /* a persistent grid: what does not fit the chip would only queue */
    const unsigned grid = (unsigned)(blocks < resident ? blocks : resident);
    u64 *bigAt = big.as<u64>();
    if (levels == 2u) {
      DevIndex dev = g->dev;
      dev.pairSuperInLds = pairSuperInLds ? 1u : 0u;
      const size_t lds = pairSuperInLds ? (size_t)g->dev.numPairSuper * 64u : 0u;
      const u64 pairBlocks = (outLen + kSeedGroupsPerBlock - 1) / kSeedGroupsPerBlock;
      const unsigned pairGrid = (unsigned)(pairBlocks < resident ? pairBlocks : resident);
#define AWFM_PAIR_LEVEL(O, NR) \
  hipLaunchKernelGGL((deepSeedPairLevelKernel<O, NR>), dim3(pairGrid), dim3(kThreads), lds, awfmGpuSetupStream, dev, parent, len, outLen, nxt.as<ulonglong2>(), bigAt)
      if (narrow) {
        if (out == 1u) AWFM_PAIR_LEVEL(1, true);
        else AWFM_PAIR_LEVEL(0, true);
      } else if (out == 2u) AWFM_PAIR_LEVEL(2, false);
      else AWFM_PAIR_LEVEL(0, false); /* (format 1 is a narrow image's) */
#undef AWFM_PAIR_LEVEL
    } else if (g->amino) {
      const u64 aminoBlocks = (outLen + kThreads / 4 - 1) / (kThreads / 4);
      const unsigned aminoGrid = (unsigned)(aminoBlocks < resident ? aminoBlocks : resident);
      if (out == 2u)
        hipLaunchKernelGGL((aminoDeepSeedLevelKernel<2>), dim3(aminoGrid), dim3(kThreads), 0, awfmGpuSetupStream, g->dev, parent, len, outLen, nxt.as<ulonglong2>(), bigAt);
      else if (out == 1u)
        hipLaunchKernelGGL((aminoDeepSeedLevelKernel<1>), dim3(aminoGrid), dim3(kThreads), 0, awfmGpuSetupStream, g->dev, parent, len, outLen, nxt.as<ulonglong2>(), bigAt);
      else
        hipLaunchKernelGGL((aminoDeepSeedLevelKernel<0>), dim3(aminoGrid), dim3(kThreads), 0, awfmGpuSetupStream, g->dev, parent, len, outLen, nxt.as<ulonglong2>(), bigAt);
    } else if (out == 2u)
      hipLaunchKernelGGL((deepSeedLevelKernel<kUnroll, 2>), dim3(grid), dim3(kThreads), 0, awfmGpuSetupStream, g->dev, parent, len, outLen, nxt.as<ulonglong2>(), bigAt);
    else if (out == 1u)
      hipLaunchKernelGGL((deepSeedLevelKernel<kUnroll, 1>), dim3(grid), dim3(kThreads), 0, awfmGpuSetupStream, g->dev, parent, len, outLen, nxt.as<ulonglong2>(), bigAt);
    else
      hipLaunchKernelGGL((deepSeedLevelKernel<kUnroll, 0>), dim3(grid), dim3(kThreads), 0, awfmGpuSetupStream, g->dev, parent, len, outLen, nxt.as<ulonglong2>(), bigAt);
    BUILD_TRY(hipGetLastError());
    BUILD_TRY(awfmGpuSetupSync());
    clock_gettime(CLOCK_MONOTONIC, &tc);
    if (awfmKnob(AWFM_KNOB_VERBOSE))
      fprintf(stderr, "[awfm deep seed] level %u -> %u: %llu entries, allocation %.3f s, kernel %.3f s\n", L, L + levels, (unsigned long long)outLen,
              (double)(tb.tv_sec - ta.tv_sec) + 1e-9 * (double)(tb.tv_nsec - ta.tv_nsec), (double)(tc.tv_sec - tb.tv_sec) + 1e-9 * (double)(tc.tv_nsec - tb.tv_nsec));
    cur.reset();
    cur.p = nxt.release();
    curBytes = outLen * entryBytes;
    parent = cur.as<ulonglong2>();
    len = outLen;
    L += levels;
  }
  *bytesOut = len * (format ? 8 : 16);
  *tableOut = cur.release();
  if (formatOut) *formatOut = format;
  if (bigOut) *bigOut = big.release();
  return true;
}

/* awfm_device.h: the tables of the k-mer lengths 1 .. maxDepth in one allocation, level d at entry awfmLengthTableAt(d) */
bool awfmGpuBuildLengthTables(const AwFmGpuIndex *g, unsigned maxDepth, void **tableOut, uint64_t *bytesOut, void **bigOut) {
  *tableOut = nullptr;
  *bytesOut = 0;
  *bigOut = nullptr;
  /* the entries follow the deeper table's format (DevIndex::deepNarrow): {sp, length} beside format 1, sp36 | length28 beside
   * format 2 */
  if (g->amino || maxDepth < 1u || maxDepth > 15u || (g->dev.deepNarrow != 1u && g->dev.deepNarrow != 2u)) {
    awfmGpuSetError("length tables: nucleotide images with an 8-byte deeper table, k-mer lengths 1..15");
    return false;
  }
  const bool wide = g->dev.deepNarrow == 2u;
  DeviceGuard guard(g->device);
  const u64 entries = awfmLengthTableAt(maxDepth + 1u);
  DeviceBuffer table, big;
  if (!table.alloc(entries * sizeof(uint2))) return false;
  if (wide) {
    if (!big.alloc(15u * kLengthWideBigStride * 8u)) return false;
    BUILD_TRY(hipMemset(big.p, 0, 15u * kLengthWideBigStride * 8u));
  }
  uint2 *base = table.as<uint2>();
  u64 *bigAt = big.as<u64>();
  if (wide) hipLaunchKernelGGL(lengthLettersKernel<true>, dim3(1), dim3(64), 0, 0, g->dev, base, bigAt);
  else hipLaunchKernelGGL(lengthLettersKernel<false>, dim3(1), dim3(64), 0, 0, g->dev, base, bigAt);
  BUILD_TRY(hipGetLastError());
  const u64 resident = (u64)g->numCUs * 8u;
  u64 len = 4;
  for (unsigned d = 1; d < maxDepth; d++, len *= 4u) { /* level d + 1 */
    const u64 outLen = len * 4u;
    uint2 *out = base + awfmLengthTableAt(d + 1u);
    if (d + 1u == g->dev.seedK && g->dev.seed) {
      const u64 blocks = (outLen + 255u) / 256u;
      const unsigned grid = (unsigned)(blocks < resident ? blocks : resident);
      if (wide) hipLaunchKernelGGL(lengthFromSeedKernel<true>, dim3(grid), dim3(256), 0, 0, g->dev.seed, outLen, out, bigAt, d + 1u);
      else hipLaunchKernelGGL(lengthFromSeedKernel<false>, dim3(grid), dim3(256), 0, 0, g->dev.seed, outLen, out, bigAt, d + 1u);
    } else {
      const u64 blocks = (outLen + kSeedGroupsPerBlock - 1) / kSeedGroupsPerBlock;
      const unsigned grid = (unsigned)(blocks < resident ? blocks : resident);
      const uint2 *parent = (const uint2 *)(base + awfmLengthTableAt(d));
      if (wide) hipLaunchKernelGGL(lengthLevelKernel<true>, dim3(grid), dim3(kThreads), 0, 0, g->dev, parent, len, outLen, out, bigAt, d + 1u);
      else hipLaunchKernelGGL(lengthLevelKernel<false>, dim3(grid), dim3(kThreads), 0, 0, g->dev, parent, len, outLen, out, bigAt, d + 1u);
    }
    BUILD_TRY(hipGetLastError());
  }
  BUILD_TRY(hipDeviceSynchronize());
  *bytesOut = entries * sizeof(uint2);
  *tableOut = table.release();
  *bigOut = big.release();
  return true;
}

extern "C" enum AwFmReturnCode awfmGpuCreateIndex(struct AwFmIndex **index, const struct AwFmIndexConfiguration *config,
                                                  const uint8_t *sequence, uint64_t sequenceLength,
                                                  int sequenceOnDevice, const char *fileSrc, int device) {
  return awfmGpuCreateIndexWithFasta(index, config, sequence, sequenceLength, sequenceOnDevice, fileSrc, device, nullptr);
}

/* fastaVector (may be NULL): record table of an index built from FASTA; owned by the index on success, left to
 * the caller on failure (the contract of awfmCreateIndexWithFasta) */
extern "C" enum AwFmReturnCode awfmGpuCreateIndexWithFasta(struct AwFmIndex **index,
                                                           const struct AwFmIndexConfiguration *config,
                                                           const uint8_t *sequence, uint64_t sequenceLength,
                                                           int sequenceOnDevice, const char *fileSrc, int device,
                                                           struct FastaVector *fastaVector) {
  if (!index || !config || !sequence) {
    awfmGpuSetError("awfmGpuCreateIndex: null argument");
    return AwFmNullPtrError;
  }
  *index = nullptr;
  if (awfmGpuDeviceCount() <= 0) {
    awfmGpuSetError("awfmGpuCreateIndex: no HIP device available");
    return AwFmGeneralFailure;
  }
  const u64 n = sequenceLength + 1; /* bwtLength */
  if (config->suffixArrayCompressionRatio == 0) {
    awfmGpuSetError("awfmGpuCreateIndex: suffixArrayCompressionRatio must be >= 1");
    return AwFmGeneralFailure;
  }
  if (device < 0) {
    const char *env = awfmKnob(AWFM_KNOB_DEVICE);
    if (env && *env)
      device = atoi(env);
    else if (hipGetDevice(&device) != hipSuccess)
      device = 0;
  }
  DeviceGuard guard(device);
  if (!guard.ok) {
    awfmGpuSetError("awfmGpuCreateIndex: hipSetDevice failed");
    return AwFmGeneralFailure;
  }
  const bool verbose = awfmKnob(AWFM_KNOB_VERBOSE) != nullptr;
  const bool amino = config->alphabetType == AwFmAlphabetAmino;
  /* 32-bit suffix positions and ranks while they fit ($AWFM_GPU_DIAG build_wide=1: 64-bit ones on any text, for tests) */
  const char *wideEnv = awfmGpuDiag("build_wide");
  const bool wide = n > 0xFFFFFFFEull || (wideEnv && *wideEnv && *wideEnv != '0');
  const u64 numBlocks = awfmNumBlocks(n);
  const unsigned refWords = amino ? 44 : 20;
  const unsigned letters = amino ? 22 : 6;
  const unsigned card = amino ? 20 : 4;
  const unsigned K = config->kmerLengthInSeedTable;
  const u64 seedLen = awfmKmerTableLength(config->alphabetType, K);

  struct AwFmIndex *ix = awfmIndexAlloc(config, n);
  if (!ix) {
    awfmGpuSetError("awfmGpuCreateIndex: host allocation failed");
    return AwFmAllocationFailure;
  }
  ix->versionNumber = AWFM_VERSION_NUMBER;
  ix->featureFlags = fastaVector ? (1u << AWFM_FEATURE_BIT_FASTA_VECTOR) : 0u;
  auto failWith = [&](enum AwFmReturnCode rc) {
    awFmDeallocIndex(ix); /* ix->fastaVector is still NULL: the caller keeps the record table */
    return rc;
  };
#define STEP(expr)                                \
  do {                                            \
    if (!(expr)) return failWith(AwFmGeneralFailure); \
  } while (0)
#define STEP_HIP(call)                            \
  do {                                            \
    hipError_t e__ = (call);                      \
    if (e__ != hipSuccess) {                      \
      awfmGpuSetHipError(#call, e__);             \
      return failWith(AwFmGeneralFailure);        \
    }                                             \
  } while (0)

  /* 1. text */
  DeviceBuffer dText, dRaw;
  STEP(dText.alloc(n + 64));
  const unsigned char *rawDev = sequence;
  if (!sequenceOnDevice) {
    STEP(dRaw.alloc(sequenceLength));
    STEP_HIP(hipMemcpy(dRaw.p, sequence, sequenceLength, hipMemcpyHostToDevice));
    rawDev = dRaw.as<unsigned char>();
  }
  hipLaunchKernelGGL(sanitizeKernel, dim3(gridOf(n)), dim3(256), 0, 0, rawDev, (u64)sequenceLength, amino ? 1 : 0,
                     dText.as<unsigned char>());
  STEP_HIP(hipGetLastError());
  dRaw.reset();

  /* 2. suffix array */
  DeviceBuffer dSa;
  STEP(dSa.alloc(n * (wide ? 8 : 4)));
  STEP(wide ? buildSuffixArray<u64>(dText.as<unsigned char>(), n, dSa.as<u64>(), verbose)
            : buildSuffixArray<u32>(dText.as<unsigned char>(), n, dSa.as<u32>(), verbose));

  /* 3. reference-layout blocks */
  DeviceBuffer dRef, dCounts, dScanned, dSentinel, temp;
  size_t tempBytes = 0;
  STEP(dRef.alloc(numBlocks * refWords * 8));
  STEP(dCounts.alloc((u64)letters * numBlocks * 8));
  STEP(dScanned.alloc((u64)letters * numBlocks * 8));
  STEP(dSentinel.alloc(8));
  STEP_HIP(hipMemset(dSentinel.p, 0, 8));
  STEP(wide ? launchBwtBlocks<u64>(amino, dText.as<unsigned char>(), dSa.p, n, numBlocks, dRef.as<u64>(), dCounts.as<u64>(),
                                   dSentinel.as<u64>())
            : launchBwtBlocks<u32>(amino, dText.as<unsigned char>(), dSa.p, n, numBlocks, dRef.as<u64>(), dCounts.as<u64>(),
                                   dSentinel.as<u64>()));
  u64 totals[24] = {0};
  for (unsigned l = 0; l < letters; l++) {
    STEP(exclusiveSum<u64>(dCounts.as<u64>() + (u64)l * numBlocks, dScanned.as<u64>() + (u64)l * numBlocks, numBlocks, temp,
                         tempBytes));
    u64 lastScan = 0, lastCount = 0;
    STEP_HIP(hipMemcpy(&lastScan, dScanned.as<u64>() + (u64)l * numBlocks + (numBlocks - 1), 8, hipMemcpyDeviceToHost));
    STEP_HIP(hipMemcpy(&lastCount, dCounts.as<u64>() + (u64)l * numBlocks + (numBlocks - 1), 8, hipMemcpyDeviceToHost));
    totals[l] = lastScan + lastCount;
  }
  {
    const u64 threads = numBlocks * (amino ? 24 : 8);
    if (amino)
      hipLaunchKernelGGL(baseOccurrenceKernel<true>, dim3(gridOf(threads)), dim3(256), 0, 0, dScanned.as<u64>(), numBlocks,
                         dRef.as<u64>());
    else
      hipLaunchKernelGGL(baseOccurrenceKernel<false>, dim3(gridOf(threads)), dim3(256), 0, 0, dScanned.as<u64>(),
                         numBlocks, dRef.as<u64>());
    STEP_HIP(hipGetLastError());
  }
  dCounts.reset();
  dScanned.reset();
  dText.reset();
  u64 sentinelPos = 0;
  STEP_HIP(hipMemcpy(&sentinelPos, dSentinel.p, 8, hipMemcpyDeviceToHost));
  /* prefix sums (ref src/AwFmCreate.c:338-344) */
  ix->prefixSums[0] = 1;
  for (unsigned i = 1; i < card + 2; i++) ix->prefixSums[i] = ix->prefixSums[i - 1] + totals[i - 1];

  /* 4. device image: blocks + superblock bases, prefix sums, seed table, packed SA */
  DeviceBuffer dBlocks, dSuper, dPrefix, dSeedA, dSeedB, dPacked;
  const unsigned superShift = awfmSuperShift(amino, n);
  if (!amino && awfmNumSuper(n, false, superShift) > kMaxNucSuper) {
    awfmGpuSetError("awfmGpuCreateIndex: nucleotide device images hold at most 64 superblocks");
    return failWith(AwFmUnsupportedVersionError);
  }
  STEP(dBlocks.alloc(awfmDeviceBlocks(n) * awfmDeviceBlockBytes(amino)));
  STEP(dSuper.alloc(awfmSuperBytes(n, amino, superShift)));
  {
    unsigned long long again = 0; /* the sentinel's position is known from the BWT pass already */
    STEP(awfmGpuRelayout(dRef.p, n, amino, superShift, dBlocks.p, dSuper.p, &again));
  }
  {
    u64 prefix[24] = {0};
    memcpy(prefix, ix->prefixSums, (card + 2) * sizeof(u64));
    STEP(dPrefix.alloc(sizeof prefix));
    STEP_HIP(hipMemcpy(dPrefix.p, prefix, sizeof prefix, hipMemcpyHostToDevice));
  }
  DevIndex dev{};
  dev.blocks = dBlocks.as<uint4>();
  dev.super = dSuper.as<u64>();
  dev.numSuper = (unsigned)awfmNumSuper(n, amino, superShift);
  dev.nucSuperShift = superShift;
  dev.prefixSums = dPrefix.as<u64>();
  dev.bwtLength = n;
  dev.sentinelPos = sentinelPos;
  dev.seedK = K;
  STEP(dSeedA.alloc(seedLen * 16));
  if (K == 0) {
    STEP_HIP(hipMemset(dSeedA.p, 0, 16));
  } else {
    STEP(dSeedB.alloc(seedLen * 16));
    std::vector<u64> level1(2 * card);
    for (unsigned a = 0; a < card; a++) {
      level1[2 * a] = ix->prefixSums[a];
      level1[2 * a + 1] = ix->prefixSums[a + 1] - 1;
    }
    /* ping-pong so that the last level lands in dSeedA */
    DeviceBuffer *cur = (K % 2 == 1) ? &dSeedA : &dSeedB;
    DeviceBuffer *nxt = (K % 2 == 1) ? &dSeedB : &dSeedA;
    STEP_HIP(hipMemcpy(cur->p, level1.data(), level1.size() * 8, hipMemcpyHostToDevice));
    u64 len = card;
    for (unsigned L = 1; L < K; L++) {
      const u64 outLen = len * card;
      const u64 blocks = (outLen + kSeedGroupsPerBlock - 1) / kSeedGroupsPerBlock;
      const unsigned grid = (unsigned)(blocks < 2048 * 4 ? blocks : 2048 * 4);
      if (amino)
        hipLaunchKernelGGL(seedLevelKernel<true>, dim3(grid), dim3(kThreads), 0, 0, dev, cur->as<ulonglong2>(), len, outLen,
                           nxt->as<ulonglong2>());
      else
        hipLaunchKernelGGL(seedLevelKernel<false>, dim3(grid), dim3(kThreads), 0, 0, dev, cur->as<ulonglong2>(), len,
                           outLen, nxt->as<ulonglong2>());
      STEP_HIP(hipGetLastError());
      std::swap(cur, nxt);
      len = outLen;
    }
    STEP_HIP(hipDeviceSynchronize());
    if (cur != &dSeedA) { /* cannot happen with the parity choice above, kept as a guard */
      awfmGpuSetError("awfmGpuCreateIndex: seed table ping-pong parity");
      return failWith(AwFmGeneralFailure);
    }
    dSeedB.reset();
  }
  ix->suffixArray.valueBitWidth = awfmSaWidth(n);
  ix->suffixArray.compressedByteLength = awfmSaPackedBytes(n, config->suffixArrayCompressionRatio);
  const u64 saWords = (ix->suffixArray.compressedByteLength + 15) / 16 * 2 + 32; /* +256 B: 128-byte window reads */
  STEP(dPacked.alloc(saWords * 8));
  {
    const u64 samples = awfmSaSampleCount(n, config->suffixArrayCompressionRatio);
    const unsigned ratio = (unsigned)config->suffixArrayCompressionRatio, width = (unsigned)ix->suffixArray.valueBitWidth;
    STEP(wide ? launchPackSampledSa<u64>(dSa.p, samples, ratio, width, saWords, dPacked.as<u64>())
              : launchPackSampledSa<u32>(dSa.p, samples, ratio, width, saWords, dPacked.as<u64>()));
  }
  STEP_HIP(hipDeviceSynchronize());
  /* the array is kept for the image this index gets below, which takes it as its full suffix array: 32-bit positions as they
   * are, 64-bit ones as 40-bit entries (round 6: images of 2^32 positions and more locate through the full array too) */
  bool stashWide = false;
  if (wide && config->suffixArrayCompressionRatio > 1 && n < (1ull << 40)) {
    DeviceBuffer dDense40;
    if (dDense40.alloc(awfmDenseSaBytes(n, true))) {
      hipLaunchKernelGGL((packDense40Kernel<u64>), dim3(2048), dim3(256), 0, 0, (const u64 *)dSa.p, n, dDense40.as<unsigned>());
      STEP_HIP(hipGetLastError());
      STEP_HIP(hipDeviceSynchronize());
      dSa.reset();
      dSa.p = dDense40.release();
      stashWide = true;
    } else {
      (void)hipGetLastError();
      dSa.reset();
    }
  } else if (wide) {
    dSa.reset();
  }

  /* 5. download the reference-layout arrays */
  STEP_HIP(hipMemcpy(ix->bwtBlockList.asNucleotide, dRef.p, numBlocks * refWords * 8, hipMemcpyDeviceToHost));
  dRef.reset();
  STEP_HIP(hipMemcpy(ix->kmerSeedTable, dSeedA.p, seedLen * 16, hipMemcpyDeviceToHost));
  ix->suffixArray.values = (uint8_t *)malloc(ix->suffixArray.compressedByteLength);
  if (!ix->suffixArray.values) {
    awfmGpuSetError("awfmGpuCreateIndex: host allocation failed");
    return failWith(AwFmAllocationFailure);
  }
  STEP_HIP(hipMemcpy(ix->suffixArray.values, dPacked.p, ix->suffixArray.compressedByteLength, hipMemcpyDeviceToHost));
  ix->suffixArrayFileOffset = awfmSuffixArrayFileOffset(ix);
  ix->sequenceFileOffset = awfmSequenceFileOffset(ix);

  /* keep the device image for awFmParallelSearch* */
  const uint64_t deviceBytes = awfmDeviceBlocks(n) * awfmDeviceBlockBytes(amino) + awfmSuperBytes(n, amino, superShift) + seedLen * 16 + saWords * 8;
  if (dSa.p && config->suffixArrayCompressionRatio > 1) { /* hand-over: see awfm_device.h */
    awfmGpuDenseSaStash = dSa.release();
    awfmGpuDenseSaStashLength = n;
    awfmGpuDenseSaStashWide = stashWide;
  } else {
    dSa.reset();
  }
  AwFmGpuIndex *g = awfmGpuIndexAdopt(ix, device, dBlocks.release(), dSuper.release(), superShift, dSeedA.release(), dPacked.release(),
                                      dPrefix.release(), sentinelPos, deviceBytes);
  if (awfmGpuDenseSaStash) { /* the image did not want a full suffix array */
    (void)hipFree(awfmGpuDenseSaStash);
    awfmGpuDenseSaStash = nullptr;
    awfmGpuDenseSaStashLength = 0;
  }
  awfmGpuIndexRegister(ix, g);

  enum AwFmReturnCode rc = AwFmFileWriteOkay;
  ix->fastaVector = fastaVector; /* the trailer of the file is written from it */
  if (fileSrc) {
    if (config->storeOriginalSequence && sequenceOnDevice) {
      std::vector<uint8_t> hostSeq(sequenceLength ? sequenceLength : 1);
      STEP_HIP(hipMemcpy(hostSeq.data(), sequence, sequenceLength, hipMemcpyDeviceToHost));
      rc = awFmWriteIndexToFile(ix, hostSeq.data(), sequenceLength, fileSrc);
    } else {
      static const uint8_t none = 0;
      rc = awFmWriteIndexToFile(ix, sequenceOnDevice ? &none : sequence, sequenceLength, fileSrc);
    }
  }
  if (awFmReturnCodeIsFailure(rc)) ix->fastaVector = nullptr; /* the caller keeps ownership on failure */
  if (!config->keepSuffixArrayInMemory && fileSrc) { /* ref src/AwFmCreate.c:128-131 */
    free(ix->suffixArray.values);
    ix->suffixArray.values = nullptr;
  }
#undef STEP
#undef STEP_HIP
  *index = ix;
  return rc;
}
