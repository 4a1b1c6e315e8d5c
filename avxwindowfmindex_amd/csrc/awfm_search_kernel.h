/*
 * awfm_search_kernel.h -- the batched backward-search kernel (seed lookup + extension).
 *
 * One query is owned by a group of G lanes (G = 4, 2 or 1; amino 4 or 2).  A device block is 4 slices of 32
 * positions (awfm_device.h); lane j of the group loads slices j*S .. j*S+S-1 (S = 4/G) -- contiguous bytes, so the
 * group always reads a whole 64-B (nucleotide) or 128-B (amino) block --, ranks its own 32*S positions with bit
 * operations + popcount, and the partial counts are summed over the group with DPP adds.  Smaller G means more
 * queries per wave (64/G) -- more dependent chains in flight and fewer redundant per-query
 * instructions per lane -- at the price of S load instructions per block instead of one.
 *
 * The last 32 characters of a k-mer live in registers (W = 8/G aligned dwords per lane +
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
#include "awfm_pair.h"

namespace {

/* length of a query from its {start, end} offsets; uses all four loaded dwords, so no register of the
 * pair is dead (and re-used, which would need a wait) while the load is still in flight */
__device__ __forceinline__ unsigned pairLength(const ulonglong2 &o) {
  const unsigned long long d = o.y - o.x;
  return d > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)d;
}

/*
 * INDIRECT: the kernel searches only the queries listed in the tail of an ordered record array (the ones the
 * ordered path, awfm_ordered_kernel.h, leaves to this kernel): record i of the tail is record
 * subsetTotal - *subsetCount + i of `subset` (records of subsetStride bytes), its first 32-bit word at byte
 * subsetIndexAt the query number.
 */
/*
 * PAIR (nucleotide, G = 4, hits-only callers): two characters per block read through the pair image (awfm_pair.h)
 * wherever the next two characters are a,c,g,t inside the register window.  A k-mer without hits may then end in a
 * different empty range than the letter-by-letter stepping ends in, which the hits-only contract allows
 * (awfmGpuSearchHits); k-mers with hits get exactly the range of the single steps.  The memory system moves whole 128-B
 * lines whatever a block's size, so a pair block -- one line, two steps -- halves the lines of a search.
 */
template <bool AMINO, int G, bool CSR, bool TALLY, bool NARROW, bool INDIRECT = false, bool PAIR = false>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(G >= 4 && !TALLY && !CSR && !AMINO && !INDIRECT && !PAIR ? (NARROW ? 8 : 6) : 2, 8)))
    searchKernel(const DevIndex ix, const unsigned char *__restrict__ chars,
                 const unsigned long long *__restrict__ offsets, const unsigned fixedLength,
                 const unsigned long long numQueries, ulonglong2 *__restrict__ ranges, unsigned *__restrict__ counts,
                 unsigned long long *__restrict__ tally, const unsigned char *__restrict__ subset = nullptr,
                 const unsigned subsetStride = 0, const unsigned subsetIndexAt = 0,
                 const unsigned long long subsetTotal = 0, const unsigned *__restrict__ subsetCount = nullptr,
                 const SparseOut sparse = SparseOut(),
                 /* launched beside a lookup-first kernel whose sample decides on the device which of the two works */
                 const unsigned *__restrict__ skipWhenLookup = nullptr, const unsigned skipSamples = 0) {
  if (skipWhenLookup && lookupChosen(skipWhenLookup, skipSamples, false)) return; /* uniform */
  if (INDIRECT && *subsetCount == 0u) return; /* nothing was left to this kernel (the usual case): no table is staged */
  constexpr int W = 8 / G; /* window dwords per lane: the group holds the last 32 characters of its k-mer */
  constexpr int S = (int)kSlices / G; /* block slices per lane */
  constexpr int kGroups = kThreads / G;
  typedef typename PositionType<NARROW>::type pos_t;
  __shared__ unsigned long long sC[24];
  __shared__ unsigned sPow[32];
  __shared__ unsigned sPowDeep[AMINO ? 32 : 1]; /* amino: the same weights for the device-only deeper table */
  __shared__ AminoShared sAmino;
  __shared__ unsigned sMask[(kBlockMask + 1) * kSlices]; /* sMask[local * 4 + slice] = bits of the slice at positions <= local */
  __shared__ unsigned long long sSuper[!AMINO && !NARROW ? kMaxNucSuper * 4 : 1];
  __shared__ unsigned long long sPairC[PAIR ? 16 : 1];
  static_assert(!PAIR || (!AMINO && G == 4 && !TALLY), "pair steps: nucleotide, 4 lanes per query, not the tally");
  extern __shared__ unsigned sPairSuper[]; /* PAIR with ix.pairSuperInLds: the 16 pair bases of every superblock */
  if (PAIR) pairStageTables<NARROW, 16u>(ix, sPairC, sPairSuper);
  const unsigned card = AMINO ? 20u : 4u;
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  stageMaskTable(sMask);
  if (!AMINO) nucStageSuper<NARROW>(ix, sSuper);
  if (threadIdx.x < 32) {
    /* weight of seed character j: card^(k-1-j) (ref src/AwFmKmerTable.c:26-32) */
    unsigned w = 1;
    for (unsigned e = threadIdx.x + 1; e < ix.seedK; e++) w *= card;
    sPow[threadIdx.x] = w;
    if (AMINO) {
      unsigned d = 1;
      for (unsigned e = threadIdx.x + 1; e < ix.deepK; e++) d *= card;
      sPowDeep[threadIdx.x] = d;
    }
  }
  if (AMINO) aminoStageTables(sAmino);
  __syncthreads();

  const unsigned lane = threadIdx.x & 63u;
  const unsigned gl = threadIdx.x % G; /* lane within the group */
  const unsigned firstWord = gl * W;   /* first of this lane's window dwords */
  const unsigned firstSlice = gl * S;  /* first of this lane's block slices */
  const unsigned long long numGroups = (unsigned long long)gridDim.x * kGroups;
  const unsigned long long groupId = ((unsigned long long)blockIdx.x * kThreads + threadIdx.x) / G;
  const unsigned K = ix.seedK;
  const unsigned long long charsEnd = CSR ? offsets[numQueries] : numQueries * (unsigned long long)fixedLength;
  /* window loads are aligned dwords; offsets are clamped to the last dword that still holds a query byte */
  const unsigned charsMisalign = (unsigned)((unsigned long long)chars & 3ull);
  const unsigned char *charsAligned = chars - charsMisalign; /* stays a global-address-space pointer */
  const unsigned long long charsLast = (charsEnd + charsMisalign - 1ull) & ~3ull;

  unsigned long long tSeeded = 0, tSteps = 0, tBlocks = 0, tChars = 0;

  /* prefetched query: raw {start,end} offsets and raw window dwords (W+1 aligned dwords cover 4*W bytes) */
  ulonglong2 nOff = make_ulonglong2(0ull, 0ull);
  unsigned nRaw[W + 1];
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
    return o.x + wb + 4u * firstWord + charsMisalign;
  };
  auto requestWindow = [&](const ulonglong2 &o) {
    const unsigned long long first = windowStart(o) & ~3ull;
#pragma unroll
    for (int w = 0; w <= W; w++) {
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
  for (int w = 0; w <= W; w++) nRaw[w] = 0u;
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
    unsigned win[W];
    {
      const unsigned shift = (unsigned)windowStart(nOff) & 3u;
#pragma unroll
      for (int w = 0; w < W; w++) win[w] = __builtin_amdgcn_alignbyte(nRaw[w + 1], nRaw[w], shift);
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
      const unsigned word = rel >> 2;      /* 0..7: lane word/W, register word%W */
      unsigned mine = win[0];
      if (W == 2) { /* named copies: hipcc turns the select over a two-element array back into an indexed (scratch) load */
        unsigned lo = win[0], hi = win[W - 1];
        asm volatile("" : "+v"(lo), "+v"(hi));
        mine = (word & 1u) ? hi : lo;
      } else {
#pragma unroll
        for (int w = 1; w < W; w++) mine = (word % W) == (unsigned)w ? win[w] : mine;
      }
      const unsigned v = groupShfl<G>(mine, word / W);
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
        unsigned partial = 0, partialDeep = 0;
        bool ambiguous = false, ambiguousDeep = false;
        const unsigned DK = ix.deepK; /* device-only deeper table (0: none): the index over the last DK characters */
        const bool tryDeep = DK != 0u && len >= DK;
        if (tryTable) {
#pragma unroll
          for (int w = 0; w < W; w++) {
#pragma unroll
            for (unsigned b = 0; b < 4; b++) {
              const unsigned i = wb + 4u * (firstWord + w) + b; /* index in the query */
              const int j = (int)i - (int)(len - K);             /* index in the seed */
              const unsigned c = (win[w] >> (8u * b)) & 0xFFu;
              const bool inSeed = i < len && j >= 0;
              const unsigned letter = aminoLetterIndex(sAmino, c);
              ambiguous |= inSeed && aminoIsAmbiguous(c);
              partial += inSeed ? letter * sPow[j & 31] : 0u;
              const int jd = (int)i - (int)(len - DK);           /* index in the last DK characters */
              const bool inDeep = tryDeep && i < len && jd >= 0;
              /* (any character that is not one of the 20 letters: its index 20 would carry into the next digit of the
               * sum and name another k-mer's entry; such k-mers start from the index's own table, as in the reference) */
              ambiguousDeep |= inDeep && (aminoIsAmbiguous(c) || letter >= 20u);
              partialDeep += inDeep ? letter * sPowDeep[jd & 31] : 0u;
            }
          }
        }
        index = groupSum<G>(partial);
        const unsigned long long ballot = __ballot(ambiguous);
        ambiguousGroup = ((unsigned)(ballot >> (lane & ~(unsigned)(G - 1))) & ((1u << G) - 1u)) != 0u;
        if (tryDeep) { /* (uniform across the group) same answer as the seed entry followed by DK - K extension steps */
          const unsigned indexDeep = groupSum<G>(partialDeep);
          const unsigned long long ballotDeep = __ballot(ambiguousDeep);
          if (((unsigned)(ballotDeep >> (lane & ~(unsigned)(G - 1))) & ((1u << G) - 1u)) == 0u) {
            const ulonglong2 r = aminoDeepSeedEntry(ix, indexDeep);
            sp = (pos_t)r.x;
            ep = (pos_t)r.y;
            pos = (int)(len - DK) - 1;
            deep = true;
          }
        }
      } else {
        /* Nucleotide: with 4 letters the table index is simply the 2-bit letter codes of the last K
         * characters concatenated, first character most significant.  Decode the window 4 characters at a
         * time (SWAR): for a,c,g,t,u the bits (c>>1)&3 are 0,1,3,2,2 and x^(x>>1) maps them to 0,1,2,3,3;
         * a byte is a valid letter iff re-encoding its code gives the byte back (u matches t up to bit 0). */
        unsigned long long codes = 0; /* this lane's 4*W characters, 2 bits each, first character on top */
        unsigned bad = 0;             /* bit i: character i of this lane's part is not a,c,g,t,u */
#pragma unroll
        for (int w = 0; w < W; w++) {
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
        const unsigned long long allCodes = groupSum64<G>(codes << (64 - 8 * W * ((int)gl + 1)));
        const unsigned allBad = groupSum<G>(bad << (4 * W * gl));
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
            const ulonglong2 r = deepSeedEntry(ix, tail & ((1ull << (2u * DK)) - 1ull));
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
      if (TALLY) { /* the accounting is in the reference's 256-position blocks, whatever the device layout */
        tSteps++;
        tBlocks += ((unsigned long long)(sp - 1) >> 8) == ((unsigned long long)ep >> 8) ? 1ull : 2ull;
      }
      const bool pairStep = PAIR && pos > (int)wb && (badTop >> 30) == 0u;
      if (pairStep) {
        /* ---- two characters, both a,c,g,t/u inside the register window: one pair block ---- */
        const unsigned c2 = (unsigned)rem & 3u, c1 = (unsigned)(rem >> 2) & 3u;
        /* exact: a k-mer that dies inside the pair ends in the range the letter-by-letter stepping ends in */
        const PairStep did = pairSearchStep<NARROW, true>(ix, sPairC, sPairSuper, sMask, gl, c1 * 4u + c2, sp, ep, sC);
        if (did == kPairFlagged) nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, c2, sp, ep);
        if (did != kPairStepped && sp <= ep) nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, c1, sp, ep);
        pos--; /* the second character: the common tail below steps past the first */
        rem >>= 2;
        badTop <<= 1;
      } else
      if (!AMINO && __builtin_expect(pos >= (int)wb && (int)badTop >= 0, 1)) {
        /* ---- fast step: a,c,g,t/u inside the register window (letter from the 2-bit codes of the seed decode) ---- */
        nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, (unsigned)rem & 3u, sp, ep);
      } else {
        /* ---- any letter: ambiguity characters, characters before the 32-character window, the amino alphabet ---- */
        unsigned c;
        if (__builtin_expect((unsigned)pos >= wb, 1))
          c = windowChar((unsigned)pos);
        else
          c = chars[base + (unsigned)pos]; /* k-mers longer than the 32-character window */
        if (AMINO)
          aminoStepAny<G, NARROW>(ix, sC, sAmino, sMask, firstSlice, aminoLetterIndex(sAmino, c), sp, ep);
        else
          nucStepAny<G, NARROW>(ix, sC, sSuper, firstSlice, nucLetterIndex(c), sp, ep);
      }
      pos--;
      rem >>= 2;
      badTop <<= 1;
    }

    if (sparse.count) { /* kernel argument: uniform (the tail of a sparse ordered search: awfmGpuSearchHitsCompact) */
      sparseAppend(sparse, gl == 0 && sp <= ep, (unsigned)queryNumber(q), (unsigned long long)sp, (unsigned long long)ep);
    } else if (sparse.kmers) { /* the tail of a search with results in search order: the listed k-mers are the order's last */
      if (gl == 0) {
        const unsigned long long slot = subsetTotal - listed + q;
        sparse.kmers[slot] = (unsigned)queryNumber(q);
        sparse.ranges[slot] = make_ulonglong2((unsigned long long)sp, (unsigned long long)ep);
        if (counts) counts[slot] = sp <= ep ? (unsigned)(ep - sp + (pos_t)1) : 0u; /* (counts in search order too) */
      }
    } else if (gl == 0) {
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
