/*
 * awfm_search_kernel.h -- the batched backward-search kernel (seed lookup + extension).
 *
 * One query is owned by a group of G lanes (G = 8, 4, 2 or 1).  A BWT block of the device image is
 * 8 pieces (awfm_device.h); lane j of the group loads pieces j*S .. j*S+S-1 (S = 8/G, 16*S contiguous
 * bytes, so the group always reads whole 128-B lines), ranks its own 32*S positions with XOR/OR/AND +
 * popcount, and the partial counts are summed over the group with DPP adds.  Smaller G means more
 * queries per wave (64/G) -- more dependent chains in flight and fewer redundant per-query
 * instructions per lane -- at the price of S load instructions per block instead of one.
 *
 * The last 32 characters of a k-mer live in registers (4*S bytes per lane, aligned dword loads +
 * v_alignbyte), prefetched one query ahead; seed index and ambiguity test are computed from the
 * window without branches; both blocks of a step are requested before either is consumed.
 *
 * Semantics (bit-exact with the reference): ref src/AwFmParallelSearch.c:222-313,
 * src/AwFmKmerTable.c:4-51, src/AwFmSearch.c:42-159, :485-520 -- the non-seeded search over the last
 * min(len,k) characters followed by the extension loop is one right-to-left walk that stops at the
 * first invalid range and keeps it.
 */
#ifndef AWFM_SEARCH_KERNEL_H
#define AWFM_SEARCH_KERNEL_H

#include "awfm_device.h"

namespace {

/* sum over the G lanes of a group (G a power of two <= 8); every lane gets the total */
template <int G>
__device__ __forceinline__ unsigned groupSum(unsigned v) {
  if (G >= 2) v += dppMove<0xB1>(v);  /* quad_perm [1,0,3,2] */
  if (G >= 4) v += dppMove<0x4E>(v);  /* quad_perm [2,3,0,1] */
  if (G >= 8) v += dppMove<0x141>(v); /* row_half_mirror */
  return v;
}

template <int G>
__device__ __forceinline__ unsigned long long groupSum64(unsigned long long v) {
  if (G >= 2) {
    const unsigned lo = dppMove<0xB1>((unsigned)v), hi = dppMove<0xB1>((unsigned)(v >> 32));
    v += ((unsigned long long)hi << 32) | lo;
  }
  if (G >= 4) {
    const unsigned lo = dppMove<0x4E>((unsigned)v), hi = dppMove<0x4E>((unsigned)(v >> 32));
    v += ((unsigned long long)hi << 32) | lo;
  }
  if (G >= 8) {
    const unsigned lo = dppMove<0x141>((unsigned)v), hi = dppMove<0x141>((unsigned)(v >> 32));
    v += ((unsigned long long)hi << 32) | lo;
  }
  return v;
}

template <int G>
__device__ __forceinline__ unsigned groupShfl(unsigned v, unsigned srcLaneInGroup) {
  if (G == 1) return v;
  return (unsigned)__shfl((int)v, (int)srcLaneInGroup, G);
}

/* length of a query from its {start, end} offsets; uses all four loaded dwords, so no register of the
 * pair is dead (and re-used, which would need a wait) while the load is still in flight */
__device__ __forceinline__ unsigned pairLength(const ulonglong2 &o) {
  const unsigned long long d = o.y - o.x;
  return d > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)d;
}

/* 32-bit base count `slot` (0..2) of an amino piece's second half, picked with shifts (a select chain
 * on vector components makes hipcc spill the vector to LDS for dynamic indexing) */
__device__ __forceinline__ unsigned aminoCountWord(const uint4 &hi, unsigned slot) {
  const unsigned long long c01 = ((unsigned long long)hi.z << 32) | hi.y;
  const unsigned long long c2x = hi.w;
  return (unsigned)((slot == 2u ? c2x : c01) >> (slot == 1u ? 32u : 0u));
}

/* one 16-byte piece of a BWT block (default cache policy: a non-temporal load measured 25 % slower) */
__device__ __forceinline__ uint4 loadBlockPiece(const uint4 *p) { return *p; }

/* a piece as one 128-bit register tuple */
typedef unsigned Piece __attribute__((ext_vector_type(4)));

/* occurrence bits of letter code (c1,c0) in a nucleotide piece, c0m/c1m = the code bits as all-ones masks.
 * With planes x,y,z: a (00) = y&z, c (01) = x&z, g (10) = x&y, t (11) = x&~y&~z (the literals of ref
 * src/AwFmOccurrence.c:18-31), i.e. (x | a) & ((y ^ t) | c) & ((z ^ t) | g) with one v_bitop3 per factor. */
__device__ __forceinline__ unsigned nucOccFast(const Piece &pc, unsigned c0m, unsigned c1m) {
  const unsigned t0 = __builtin_amdgcn_bitop3_b32(pc.x, c0m, c1m, 0xF1); /* x | (~c0 & ~c1) */
  const unsigned t1 = __builtin_amdgcn_bitop3_b32(pc.y, c0m, c1m, 0x7C); /* (y ^ (c0 & c1)) | (c0 & ~c1) */
  const unsigned t2 = __builtin_amdgcn_bitop3_b32(pc.z, c0m, c1m, 0x7A); /* (z ^ (c0 & c1)) | (~c0 & c1) */
  return t0 & t1 & t2;
}

/* BWT positions are 32-bit when bwtLength < 2^32 (NARROW): half the integer work of the range arithmetic */
template <bool NARROW>
struct PositionType {
  typedef unsigned long long type;
};
template <>
struct PositionType<true> {
  typedef unsigned type;
};

/*
 * One backward step of a nucleotide query whose next letter is a,c,g or t/u (`letter` 0..3), by the G lanes
 * of its group.  Everything that does not depend on the block (plane selectors, position masks from the LDS
 * table sMask[local * 8 + piece], C[a]) is computed between issuing the loads and the first use of their
 * data; both blocks are requested before anything waits.  Returns whether sp-1 and ep share a block.
 * ref src/AwFmSearch.c:42-159, src/AwFmOccurrence.c:18-31, :170-217.
 */
template <int G, bool NARROW>
__device__ __forceinline__ bool nucFastStep(const DevIndex &ix, const unsigned long long *sC, const unsigned *sMask,
                                            unsigned firstPiece, unsigned letter,
                                            typename PositionType<NARROW>::type &sp,
                                            typename PositionType<NARROW>::type &ep) {
  constexpr int S = 8 / G;
  typedef typename PositionType<NARROW>::type pos_t;
  const pos_t q0 = sp - 1, q1 = ep;
  const unsigned long long blk0 = q0 >> 8, blk1 = q1 >> 8;
  const bool same = blk0 == blk1;
  Piece p0[S], p1[S];
  {
    const Piece *a0 = (const Piece *)(ix.blocks + (blk0 * 8ull + firstPiece));
#pragma unroll
    for (int s = 0; s < S; s++) p0[s] = a0[s];
  }
  /* p1 starts as "whatever the registers hold" (no instruction); lanes with one block never use it */
#pragma unroll
  for (int s = 0; s < S; s++) asm volatile("" : "=v"(p1[s]));
  if (!same) {
    const Piece *a1 = (const Piece *)(ix.blocks + (blk1 * 8ull + firstPiece));
#pragma unroll
    for (int s = 0; s < S; s++) p1[s] = a1[s];
  }
  const unsigned c0m = 0u - (letter & 1u), c1m = 0u - (letter >> 1);
  const unsigned *m0 = sMask + (((unsigned)q0 & 255u) * 8u + firstPiece);
  const unsigned *m1 = sMask + (((unsigned)q1 & 255u) * 8u + firstPiece);
  unsigned mask0[S], mask1[S];
#pragma unroll
  for (int s = 0; s < S; s++) {
    mask0[s] = m0[s];
    mask1[s] = m1[s];
  }
  const pos_t cLetter = (pos_t)sC[letter];
  const unsigned sameMask = same ? ~0u : 0u;
  unsigned n0 = 0, n1 = 0;
#pragma unroll
  for (int s = 0; s < S; s++) {
    const unsigned occ0 = nucOccFast(p0[s], c0m, c1m), occ1 = nucOccFast(p1[s], c0m, c1m);
    n0 += __popc(occ0 & mask0[s]);
    n1 += __popc(__builtin_amdgcn_bitop3_b32(occ0, occ1, sameMask, 0xE4) & mask1[s]); /* same ? occ0 : occ1 */
  }
  /* keep every loaded register allocated until here: a dead component (an unused count word) would be
   * re-used for the values above while the load is in flight, which costs a full wait before them */
#pragma unroll
  for (int s = 0; s < S; s++) asm volatile("" ::"v"(p0[s]), "v"(p1[s]));
  const unsigned kLo = 2u * letter, kHi = kLo + 1u;
  unsigned lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
#pragma unroll
  for (int s = 0; s < S; s++) {
    lo0 = (kLo % S) == (unsigned)s ? p0[s].w : lo0;
    hi0 = (kHi % S) == (unsigned)s ? p0[s].w : hi0;
    lo1 = (kLo % S) == (unsigned)s ? p1[s].w : lo1;
    hi1 = (kHi % S) == (unsigned)s ? p1[s].w : hi1;
  }
  pos_t base0, base1;
  if (NARROW) { /* counts < 2^32: the high words are zero */
    base0 = (pos_t)groupShfl<G>(lo0, kLo / S);
    base1 = (pos_t)groupShfl<G>(lo1, kLo / S);
  } else {
    base0 = (pos_t)(((unsigned long long)groupShfl<G>(hi0, kHi / S) << 32) | groupShfl<G>(lo0, kLo / S));
    base1 = (pos_t)(((unsigned long long)groupShfl<G>(hi1, kHi / S) << 32) | groupShfl<G>(lo1, kLo / S));
  }
  base1 = same ? base0 : base1;
  const unsigned packed = groupSum<G>(n0 | (n1 << 16));
  sp = cLetter + base0 + (pos_t)(packed & 0xFFFFu);
  ep = cLetter + base1 + (pos_t)(packed >> 16) - (pos_t)1;
  return same;
}

/*
 * INDIRECT: the kernel searches only the queries listed in the tail of an ordered record array (the ones the
 * ordered path, awfm_ordered_kernel.h, leaves to this kernel): record i of the tail is record
 * subsetTotal - *subsetCount + i of `subset` (records of subsetStride bytes), its first 32-bit word at byte
 * subsetIndexAt the query number.
 */
template <bool AMINO, int G, bool CSR, bool TALLY, bool NARROW, bool INDIRECT = false>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(G >= 4 && !TALLY && !CSR && !AMINO && !INDIRECT ? 8 : 2, 8)))
    searchKernel(const DevIndex ix, const unsigned char *__restrict__ chars,
                 const unsigned long long *__restrict__ offsets, const unsigned fixedLength,
                 const unsigned long long numQueries, ulonglong2 *__restrict__ ranges, unsigned *__restrict__ counts,
                 unsigned long long *__restrict__ tally, const unsigned char *__restrict__ subset = nullptr,
                 const unsigned subsetStride = 0, const unsigned subsetIndexAt = 0,
                 const unsigned long long subsetTotal = 0, const unsigned *__restrict__ subsetCount = nullptr) {
  constexpr int S = 8 / G;          /* pieces (and window dwords) per lane */
  constexpr int V = AMINO ? 2 : 1;  /* uint4 per piece */
  constexpr int kGroups = kThreads / G;
  typedef typename PositionType<NARROW>::type pos_t;
  __shared__ unsigned long long sC[24];
  __shared__ unsigned sPow[32];
  __shared__ AminoShared sAmino;
  /* nucleotide fast step: sMask[local * 8 + piece] = bits of piece `piece` at positions <= local */
  __shared__ unsigned sMask[AMINO ? 8 : 256 * 8];
  const unsigned card = AMINO ? 20u : 4u;
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  if (!AMINO) {
    for (unsigned e = threadIdx.x; e < 256u * 8u; e += kThreads) sMask[e] = sliceMask(e >> 3, e & 7u);
  }
  if (threadIdx.x < 32) {
    /* weight of seed character j: card^(k-1-j) (ref src/AwFmKmerTable.c:26-32) */
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
  const unsigned gl = threadIdx.x % G; /* lane within the group */
  const unsigned firstPiece = gl * S;
  const unsigned long long numGroups = (unsigned long long)gridDim.x * kGroups;
  const unsigned long long groupId = ((unsigned long long)blockIdx.x * kThreads + threadIdx.x) / G;
  const unsigned K = ix.seedK;
  const unsigned long long charsEnd = CSR ? offsets[numQueries] : numQueries * (unsigned long long)fixedLength;
  /* window loads are aligned dwords; offsets are clamped to the last dword that still holds a query byte */
  const unsigned charsMisalign = (unsigned)((unsigned long long)chars & 3ull);
  const unsigned char *charsAligned = chars - charsMisalign; /* stays a global-address-space pointer */
  const unsigned long long charsLast = (charsEnd + charsMisalign - 1ull) & ~3ull;

  unsigned long long tSeeded = 0, tSteps = 0, tBlocks = 0, tChars = 0;

  /* prefetched query: raw {start,end} offsets and raw window dwords (S+1 aligned dwords cover 4*S bytes) */
  ulonglong2 nOff = make_ulonglong2(0ull, 0ull);
  unsigned nRaw[S + 1];
  ulonglong2 fOff = make_ulonglong2(0ull, 0ull); /* CSR only: offsets of the query after the prefetched one */

  auto queryOffsets = [&](unsigned long long q) -> ulonglong2 {
    if (CSR) {
      const unsigned long long *o = offsets + q;
      return make_ulonglong2(o[0], o[1]);
    }
    return make_ulonglong2(q * fixedLength, q * fixedLength + fixedLength);
  };
  auto windowStart = [&](const ulonglong2 &o) -> unsigned long long { /* byte address (+misalign) of this lane's part */
    const unsigned L = pairLength(o);
    const unsigned wb = L > 32u ? L - 32u : 0u;
    return o.x + wb + 4u * firstPiece + charsMisalign;
  };
  auto requestWindow = [&](const ulonglong2 &o) {
    const unsigned long long first = windowStart(o) & ~3ull;
#pragma unroll
    for (int w = 0; w <= S; w++) {
      unsigned long long at = first + 4ull * w;
      at = at < charsLast ? at : charsLast;
      nRaw[w] = *(const unsigned *)(charsAligned + at);
    }
  };

  /* INDIRECT: i-th listed query -> query number; otherwise the identity */
  const unsigned long long listed = INDIRECT ? (unsigned long long)*subsetCount : numQueries;
  const unsigned char *subsetTail = INDIRECT ? subset + (subsetTotal - listed) * subsetStride + subsetIndexAt : nullptr;
  auto queryNumber = [&](unsigned long long i) -> unsigned long long {
    return INDIRECT ? (unsigned long long)*(const unsigned *)(subsetTail + i * subsetStride) : i;
  };
  unsigned long long q = groupId;
#pragma unroll
  for (int w = 0; w <= S; w++) nRaw[w] = 0u;
  if (q < listed) {
    nOff = queryOffsets(queryNumber(q));
    requestWindow(nOff);
  }
  if (CSR && q + numGroups < listed) fOff = queryOffsets(queryNumber(q + numGroups));

  for (; q < listed; q += numGroups) {
    /* ---- the prefetched query becomes current ---- */
    const unsigned long long base = nOff.x;
    const unsigned len = pairLength(nOff);
    const unsigned wb = len > 32u ? len - 32u : 0u;
    unsigned win[S];
    {
      const unsigned shift = (unsigned)windowStart(nOff) & 3u;
#pragma unroll
      for (int w = 0; w < S; w++) win[w] = __builtin_amdgcn_alignbyte(nRaw[w + 1], nRaw[w], shift);
    }
    if (TALLY) tChars += len;
    /* ---- prefetch the next query's window (CSR: and the offsets of the one after) ---- */
    {
      const unsigned long long qn = q + numGroups;
      if (qn < listed) {
        nOff = CSR ? fOff : queryOffsets(queryNumber(qn));
        requestWindow(nOff);
      }
      if (CSR && qn + numGroups < listed) fOff = queryOffsets(queryNumber(qn + numGroups));
    }
    /* character i (>= wb) of the query out of the register window */
    auto windowChar = [&](unsigned i) -> unsigned {
      const unsigned rel = i - wb;         /* 0..31 */
      const unsigned word = rel >> 2;      /* 0..7: lane word/S, register word%S */
      unsigned mine = win[0];
#pragma unroll
      for (int w = 1; w < S; w++) mine = (word % S) == (unsigned)w ? win[w] : mine;
      const unsigned v = groupShfl<G>(mine, word / S);
      return (v >> (8u * (rel & 3u))) & 0xFFu;
    };

    pos_t sp = 1, ep = 0;
    int pos = -1;
    unsigned long long winCodes = 0; /* nucleotide: 2-bit codes of the window, character 0 in bits 63..62 */
    unsigned winBad = 0;             /* nucleotide: bit i set when window character i is not a,c,g,t,u */
    if (len != 0) {
      /* ---- seed (ref src/AwFmKmerTable.c:4-51) ---- */
      bool seeded = false, deep = false;
      const bool tryTable = K != 0 && K <= 32u && len >= K;
      unsigned long long index = 0;
      bool ambiguousGroup = false; /* group-uniform */
      if (AMINO) {
        unsigned partial = 0;
        bool ambiguous = false;
        if (tryTable) {
#pragma unroll
          for (int w = 0; w < S; w++) {
#pragma unroll
            for (unsigned b = 0; b < 4; b++) {
              const unsigned i = wb + 4u * (firstPiece + w) + b; /* index in the query */
              const int j = (int)i - (int)(len - K);             /* index in the seed */
              const unsigned c = (win[w] >> (8u * b)) & 0xFFu;
              const bool inSeed = i < len && j >= 0;
              ambiguous |= inSeed && aminoIsAmbiguous(c);
              partial += inSeed ? aminoLetterIndex(sAmino, c) * sPow[j & 31] : 0u;
            }
          }
        }
        index = groupSum<G>(partial);
        const unsigned long long ballot = __ballot(ambiguous);
        ambiguousGroup = ((unsigned)(ballot >> (lane & ~(unsigned)(G - 1))) & ((1u << G) - 1u)) != 0u;
      } else {
        /* Nucleotide: with 4 letters the table index is simply the 2-bit letter codes of the last K
         * characters concatenated, first character most significant.  Decode the window 4 characters at a
         * time (SWAR): for a,c,g,t,u the bits (c>>1)&3 are 0,1,3,2,2 and x^(x>>1) maps them to 0,1,2,3,3;
         * a byte is a valid letter iff re-encoding its code gives the byte back (u matches t up to bit 0). */
        unsigned long long codes = 0; /* this lane's 4*S characters, 2 bits each, first character on top */
        unsigned bad = 0;             /* bit i: character i of this lane's part is not a,c,g,t,u */
#pragma unroll
        for (int w = 0; w < S; w++) {
          const unsigned word = win[w];
          unsigned t = (word >> 1) & 0x03030303u;
          t ^= (t >> 1) & 0x01010101u;
          const unsigned b0 = t & 0x01010101u, b1 = (t >> 1) & 0x01010101u, b01 = b0 & b1;
          const unsigned expect = 0x61616161u + (b0 << 1) + b1 * 6u + b01 * 11u; /* 'a','c','g','t' */
          unsigned diff = ((word | 0x20202020u) ^ expect) & ~b01;
          diff |= diff >> 4;
          diff |= diff >> 2;
          diff |= diff >> 1;
          diff &= 0x01010101u;
          const unsigned badBits = (diff & 1u) | ((diff >> 7) & 2u) | ((diff >> 14) & 4u) | ((diff >> 21) & 8u);
          const unsigned packed = ((t & 3u) << 6) | ((t >> 4) & 0x30u) | ((t >> 14) & 0x0Cu) | (t >> 24);
          codes = (codes << 8) | packed;
          bad |= badBits << (4 * w);
        }
        /* window-wide: 64-bit code string (character 0 of the window in bits 63..62) and 32-bit bad mask */
        const unsigned long long allCodes = groupSum64<G>(codes << (64 - 8 * S * ((int)gl + 1)));
        const unsigned allBad = groupSum<G>(bad << (4 * S * gl));
        winCodes = allCodes;
        winBad = allBad;
        const unsigned e = len - wb; /* characters of the query inside the window: 1..32, >= K when tryTable */
        const unsigned long long tail = e >= 32u ? allCodes : (allCodes >> (2u * (32u - e)));
        const unsigned long long kMask = K >= 32u ? ~0ull : ((1ull << (2u * K)) - 1ull);
        index = tail & kMask;
        const unsigned long long seedChars = (K >= 32u ? 0xFFFFFFFFull : ((1ull << K) - 1ull)) << (tryTable ? e - K : 0u);
        ambiguousGroup = ((unsigned long long)allBad & seedChars) != 0ull;
        /* device-only deeper table: same answer as the seed entry followed by deepK-K extension steps */
        const unsigned DK = ix.deepK;
        if (DK != 0u && len >= DK) {
          const unsigned long long deepChars = ((1ull << DK) - 1ull) << (e - DK);
          if (((unsigned long long)allBad & deepChars) == 0ull) {
            const ulonglong2 r = ix.deepSeed[tail & ((1ull << (2u * DK)) - 1ull)];
            sp = (pos_t)r.x;
            ep = (pos_t)r.y;
            pos = (int)(len - DK) - 1;
            deep = true;
          }
        }
      }
      seeded = !deep && tryTable && !ambiguousGroup && index < ix.seedLen;
      if (seeded) {
        if (TALLY) tSeeded++;
        const ulonglong2 r = ix.seed[index];
        sp = (pos_t)r.x;
        ep = (pos_t)r.y;
        pos = (int)(len - K) - 1;
      } else if (!deep) { /* ref src/AwFmSearch.c:485-502 */
        const unsigned c = windowChar(len - 1u);
        const unsigned a = AMINO ? aminoLetterIndex(sAmino, c) : nucLetterIndex(c);
        sp = (pos_t)sC[a];
        ep = (pos_t)(sC[a + 1] - 1ull);
        pos = (int)len - 2;
      }
    }

    /* ---- extension (ref src/AwFmParallelSearch.c:273-313; one step = ref src/AwFmSearch.c:42-159) ---- */
    /* nucleotide: the window's codes from the seed decode serve the extension too.  `rem` has the code of
     * character `pos` in bits 1..0 (the one before it in bits 3..2, ...), `badTop` its validity in bit 31. */
    unsigned long long rem = 0;
    unsigned badTop = 0;
    if (!AMINO) {
      const unsigned r = (unsigned)(pos - (int)wb) & 31u; /* only used while pos >= wb */
      rem = winCodes >> (62u - 2u * r);
      badTop = winBad << (31u - r);
    }
    while (pos >= 0 && sp <= ep) {
      if (!AMINO && __builtin_expect(pos >= (int)wb && (int)badTop >= 0, 1)) {
        /* ---- fast step: a,c,g,t/u inside the register window.  Everything that does not depend on
         * the block (letter, plane selectors, position masks from the LDS table, C[a]) is computed
         * between issuing the loads and the first use of their data. ---- */
        const bool same = nucFastStep<G, NARROW>(ix, sC, sMask, firstPiece, (unsigned)rem & 3u, sp, ep);
        if (TALLY) {
          tSteps++;
          tBlocks += same ? 1ull : 2ull;
        }
        pos--;
        rem >>= 2;
        badTop <<= 1;
        continue;
      }
      const pos_t q0 = sp - 1, q1 = ep;
      const unsigned long long blk0 = q0 >> 8, blk1 = q1 >> 8;
      const bool same = (q0 >> 8) == (q1 >> 8);
      uint4 p0[S][V], p1[S][V];
#pragma unroll
      for (int s = 0; s < S; s++)
#pragma unroll
        for (int v = 0; v < V; v++) {
          p0[s][v] = loadBlockPiece(ix.blocks + (blk0 * 8ull + firstPiece + s) * V + v);
          p1[s][v] = make_uint4(0u, 0u, 0u, 0u);
        }
      if (!same) {
#pragma unroll
        for (int s = 0; s < S; s++)
#pragma unroll
          for (int v = 0; v < V; v++) p1[s][v] = loadBlockPiece(ix.blocks + (blk1 * 8ull + firstPiece + s) * V + v);
      }
      unsigned c;
      if (__builtin_expect((unsigned)pos >= wb, 1))
        c = windowChar((unsigned)pos);
      else
        c = chars[base + (unsigned)pos]; /* k-mers longer than the 32-character window */
      if (TALLY) {
        tSteps++;
        tBlocks += same ? 1ull : 2ull;
      }
      const unsigned letter = AMINO ? aminoLetterIndex(sAmino, c) : nucLetterIndex(c);
      const unsigned local0 = (unsigned)q0 & 255u, local1 = (unsigned)q1 & 255u;
      unsigned n0 = 0, n1 = 0;
      pos_t base0, base1;
      if (AMINO) {
        const unsigned pm = sAmino.planeMask[letter < 24u ? letter : 23u];
        const unsigned ones = pm & 0xFFu, zeros = pm >> 8;
#pragma unroll
        for (int s = 0; s < S; s++) {
          const uint4 l1 = same ? p0[s][0] : p1[s][0], h1 = same ? p0[s][V - 1] : p1[s][V - 1];
          n0 += __popc(aminoOccSlice(p0[s][0], p0[s][V - 1], ones, zeros) & sliceMask(local0, firstPiece + s));
          n1 += __popc(aminoOccSlice(l1, h1, ones, zeros) & sliceMask(local1, firstPiece + s));
        }
        /* count of letter a: slot a%3 of piece a/3 (32-bit) */
        const unsigned piece = letter / 3u, slot = letter % 3u;
        unsigned mine0 = 0, mine1 = 0;
#pragma unroll
        for (int s = 0; s < S; s++) {
          const uint4 h0 = p0[s][V - 1], h1 = same ? p0[s][V - 1] : p1[s][V - 1];
          const unsigned w0 = aminoCountWord(h0, slot), w1 = aminoCountWord(h1, slot);
          mine0 = (piece % S) == (unsigned)s ? w0 : mine0;
          mine1 = (piece % S) == (unsigned)s ? w1 : mine1;
        }
        base0 = groupShfl<G>(mine0, piece / S);
        base1 = groupShfl<G>(mine1, piece / S);
      } else {
        const PlaneSel3 sel = nucPlaneSel(letter);
#pragma unroll
        for (int s = 0; s < S; s++) {
          const uint4 o1 = same ? p0[s][0] : p1[s][0];
          n0 += __popc(nucOccSlice(p0[s][0], sel) & sliceMask(local0, firstPiece + s));
          n1 += __popc(nucOccSlice(o1, sel) & sliceMask(local1, firstPiece + s));
        }
        if (letter < 4u) {
          /* count words 2a (low) and 2a+1 (high) of the block, i.e. piece 2a / 2a+1 */
          const unsigned kLo = 2u * letter, kHi = kLo + 1u;
          unsigned lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
#pragma unroll
          for (int s = 0; s < S; s++) {
            const unsigned w0 = p0[s][0].w, w1 = same ? p0[s][0].w : p1[s][0].w;
            lo0 = (kLo % S) == (unsigned)s ? w0 : lo0;
            hi0 = (kHi % S) == (unsigned)s ? w0 : hi0;
            lo1 = (kLo % S) == (unsigned)s ? w1 : lo1;
            hi1 = (kHi % S) == (unsigned)s ? w1 : hi1;
          }
          if (NARROW) { /* counts < 2^32: the high words are zero */
            base0 = (pos_t)groupShfl<G>(lo0, kLo / S);
            base1 = (pos_t)groupShfl<G>(lo1, kLo / S);
          } else {
            base0 = (pos_t)(((unsigned long long)groupShfl<G>(hi0, kHi / S) << 32) | groupShfl<G>(lo0, kLo / S));
            base1 = (pos_t)(((unsigned long long)groupShfl<G>(hi1, kHi / S) << 32) | groupShfl<G>(lo1, kLo / S));
          }
        } else {
          /* X: positions before the block that are not A,C,G,T or the sentinel */
          unsigned long long part0 = 0, part1 = 0;
#pragma unroll
          for (int s = 0; s < S; s++) {
            const unsigned w0 = p0[s][0].w, w1 = same ? p0[s][0].w : p1[s][0].w;
            const bool high = ((firstPiece + s) & 1u) != 0u;
            part0 += high ? ((unsigned long long)w0 << 32) : (unsigned long long)w0;
            part1 += high ? ((unsigned long long)w1 << 32) : (unsigned long long)w1;
          }
          const unsigned long long before0 = blk0 * 256ull, before1 = blk1 * 256ull;
          base0 = (pos_t)(before0 - groupSum64<G>(part0) - (ix.sentinelPos < before0 ? 1ull : 0ull));
          base1 = (pos_t)(before1 - groupSum64<G>(part1) - (ix.sentinelPos < before1 ? 1ull : 0ull));
        }
      }
      const unsigned packed = groupSum<G>(n0 | (n1 << 16));
      const pos_t cLetter = (pos_t)sC[letter];
      sp = cLetter + base0 + (pos_t)(packed & 0xFFFFu);
      ep = cLetter + base1 + (pos_t)(packed >> 16) - (pos_t)1;
      pos--;
      rem >>= 2;
      badTop <<= 1;
    }

    if (gl == 0) {
      const unsigned long long out = queryNumber(q);
      if (ranges) ranges[out] = make_ulonglong2((unsigned long long)sp, (unsigned long long)ep);
      if (counts) counts[out] = sp <= ep ? (unsigned)(ep - sp + (pos_t)1) : 0u;
    }
  }
  if (TALLY && gl == 0) { /* one lane per group carries the group's counters */
    atomicAdd(&tally[0], tSeeded);
    atomicAdd(&tally[1], tSteps);
    atomicAdd(&tally[2], tBlocks);
    atomicAdd(&tally[3], tChars);
  }
}

}  // namespace

#endif
