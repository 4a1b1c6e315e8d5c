/*
 * awfm_mixed_lookup_kernel.h -- "lookup first" for mixed-length nucleotide batches (CSR offsets; hits-only searches).
 *
 * The seed-order path takes a mixed-length batch through 16-byte records, a partition and a search kernel whose wave rounds
 * hold k-mers of every length: chains of one to fifteen dependent reads side by side, 7.6 ms per 10^8 8..30-mers at 0.39 of
 * the HBM peak -- bound by the longest chain of a round, not by bandwidth.  With one device-only table PER K-MER LENGTH below
 * the deeper table's (awfmGpuBuildLengthTables: 8-byte entries {sp, length}, 11.5 GB for the lengths 1..15) a k-mer of any
 * length is ONE table entry away from its answer or from the few steps the deeper table leaves:
 *   length <  deepK: the entry of its own length's table is its range;
 *   length == deepK: the deeper table's entry is its range;
 *   length >  deepK: the deeper table's entry, its next-step bit (two or more characters to go), then the remaining
 *                    steps -- for the few k-mers that are still alive (random 8..30-mers against 3.1 Gbp: 5 %).
 * So this kernel is lookupSearchKernel (awfm_ordered_kernel.h) with the length read per k-mer: a lane decodes four k-mers
 * (k-mer tw + 64 i + lane of the wave's 256: neighbouring lanes read neighbouring offsets and characters), has their four
 * entries in flight at once, stores the results the entries settle, and the wave takes the survivors through their steps
 * out of LDS, 16 at a time (4 lanes per k-mer: pair steps, flagged blocks and the odd step through the one-letter image).
 * What it does not cover -- k-mers with a character that is not a,c,g,t,u, no characters or more than 32 -- goes to a list
 * the general kernel searches afterwards (INDIRECT; a wave has a slot for every k-mer of its round, so no survivor does).  A sample decides, on the
 * device, between this kernel and the 16-byte-record path for the whole batch (lookupChosen).
 *
 * Round 6: the kernel exists for 32- and for 64-bit positions (NARROW), and the entries of both kinds of table are read in
 * whichever format the image built them (DevIndex::deepNarrow; lengthEntryOpen, deepSeedOpen).
 *
 * Results: every k-mer with hits gets the range the reference reaches for it (the tables hold the ranges of the reference's
 * own seed lookup and steps: see awfmGpuBuildLengthTables; ref src/AwFmSearch.c:485-520, src/AwFmKmerTable.c:4-51,
 * src/AwFmParallelSearch.c:273-313); a k-mer without hits has count 0 and an empty range (the hits-only contract of
 * awfmGpuSearchHits).
 */
#ifndef AWFM_MIXED_LOOKUP_KERNEL_H
#define AWFM_MIXED_LOOKUP_KERNEL_H

#include "awfm_ordered_kernel.h"

namespace {

constexpr unsigned kMixedSlots = 256; /* survivors a wave takes through the steps per round: as many as the round has k-mers */

/* decodeKmer (awfm_ordered_kernel.h) for a k-mer that is followed by at least 48 bytes of the character array: two 16-byte
 * loads and a dword from the aligned-down start instead of up to nine conditional dword loads, and the any-character test of
 * decodeWordAny.  bad != 0: a character that is not a,c,g,t,u. */
__device__ __forceinline__ void decodeKmerWide(const unsigned char *__restrict__ chars, unsigned long long start, unsigned len,
                                               unsigned long long &codes, unsigned &bad) {
  typedef const Dwords4 __attribute__((address_space(1))) *GlobalDwords4;
  typedef const unsigned __attribute__((address_space(1))) *GlobalWords;
  const unsigned long long at = (unsigned long long)chars + start;
  const GlobalDwords4 from = (GlobalDwords4)(at & ~3ull);
  const unsigned shift = (unsigned)at & 3u;
  const Dwords4 q0 = from[0], q1 = from[1];
  const unsigned last = ((GlobalWords)from)[8];
  const unsigned dw[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, last};
  unsigned long long c = 0;
  bad = 0;
#pragma unroll
  for (unsigned j = 0; j < 8u; j++) {
    const unsigned inKmer = len > 4u * j ? len - 4u * j : 0u; /* characters of the k-mer from this word on */
    const unsigned mask = inKmer >= 4u ? ~0u : (1u << (8u * inKmer)) - 1u;
    unsigned packed;
    decodeWordAny(__builtin_amdgcn_alignbyte(dw[j + 1u], dw[j], shift), mask, packed, bad);
    c = (c << 8) | packed;
  }
  codes = c >> (2u * (32u - len)); /* character 0 was in bits 63..62: now the last character is in bits 1..0 */
}

/* what the table entry of a k-mer says (the same reading in the kernel and in its sample); P: the position type of the
 * kernel that asks (32 bits on images below 2^32 positions, whatever the entries' format: DevIndex::deepNarrow) */
template <class P>
struct MixedVerdict {
  P sp, length;  /* the entry's range */
  bool hitNow;   /* the entry is the k-mer's range, and it is not empty */
  bool survives; /* characters to go from a range that may still hold the k-mer */
};
template <class P>
__device__ __forceinline__ MixedVerdict<P> mixedRead(const DevIndex &ix, unsigned useNext, unsigned len, unsigned long long codes, uint2 entry) {
  MixedVerdict<P> v;
  const unsigned DK = ix.deepK;
  unsigned next16 = 0xFFFFu;
  if (len == 0u) { /* not looked up: whatever was read is not an entry of this k-mer */
    v.sp = (P)1;
    v.length = (P)0;
  } else if (len >= DK) { /* the deeper table's entry */
    const ulonglong2 r = deepSeedOpen(ix, 0ull, entry, &next16);
    v.sp = (P)r.x;
    v.length = (P)(r.y + 1ull - r.x);
  } else { /* the entry of the k-mer's own length */
    const ulonglong2 r = lengthEntryOpen(ix, len, entry);
    v.sp = (P)r.x;
    v.length = (P)r.y;
  }
  v.hitNow = len != 0u && len <= DK && v.length != 0;
  /* the first step from the deeper table is a pair step when two or more characters are left and the image has its pair
   * blocks: its next-step bit says whether that step leaves anything (bit 0 of useNext: the bits are there and in use) */
  const bool bit = (useNext & 1u) == 0u || len < DK + 2u || ((next16 >> ((unsigned)(codes >> (2u * DK)) & 15u)) & 1u) != 0u;
  v.survives = len > DK && v.length != 0 && bit && (useNext & 8u) == 0u; /* (bit 3: a measurement knob that drops them) */
  return v;
}
/* where the entry of a k-mer of `len` (>= 1) characters is: entry `at` of the deeper table (len >= deepK) or of the length tables */
__device__ __forceinline__ const uint2 *mixedEntryAt(const DevIndex &ix, const uint2 *__restrict__ lengthTable,
                                                     const unsigned long long *sLevelAt, unsigned len, unsigned long long codes) {
  if (len >= ix.deepK) return (const uint2 *)ix.deepSeed + (codes & ((1ull << (2u * ix.deepK)) - 1ull));
  return lengthTable + (sLevelAt[len] + codes);
}

template <bool NARROW>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80)))
    mixedLookupSearchKernel(const DevIndex ix, const uint2 *__restrict__ lengthTable, const unsigned char *__restrict__ chars,
                            const unsigned long long *__restrict__ offsets, const unsigned long long numQueries, const unsigned useNext,
                            const unsigned *__restrict__ sampleAlive, const unsigned samples, ulonglong2 *__restrict__ ranges,
                            unsigned *__restrict__ counts, const SparseOut sparse, unsigned long long *__restrict__ leftover,
                            unsigned *__restrict__ leftoverCount, unsigned *__restrict__ keptCounters) {
  constexpr int G = 4;
  /* NARROW: 32-bit positions (awfmImageNarrow); otherwise (round 6) the 64-bit arithmetic of ref src/AwFmIndex.h:88-91 */
  typedef typename PositionType<NARROW>::type pos_t;
  __shared__ unsigned long long sC[24];
  __shared__ unsigned sMask[(kBlockMask + 1) * kSlices];
  __shared__ unsigned long long sSuper[NARROW ? 1 : kMaxNucSuper * 4];
  __shared__ unsigned long long sPairC[16];
  extern __shared__ unsigned sPairSuper[];
  __shared__ unsigned long long sLevelAt[17];
  __shared__ unsigned long long sRem[4][kMixedSlots];
  __shared__ unsigned sNum[4][kMixedSlots];
  __shared__ pos_t sSp[4][kMixedSlots], sEp[4][kMixedSlots];
  __shared__ unsigned char sLeft[4][kMixedSlots]; /* characters to go: at most 32 */
  __shared__ unsigned char sOdd[4][kMixedSlots]; /* the slots whose k-mer has one last single step to take */
  static_assert(kMixedSlots <= 256u, "slot numbers are bytes");
  constexpr unsigned kHitBuffer = 32;
  __shared__ unsigned sHitKmers[4][kHitBuffer];
  __shared__ unsigned long long sHitRanges[4][kHitBuffer][2];
  __shared__ unsigned sHitLeft[4], sWavesDone;
  if (!lookupChosen(sampleAlive, samples, true)) return; /* this batch is the 16-byte-record path's (uniform) */
  const bool PAIR = ix.pairBlocks != nullptr && (useNext & 2u) == 0u;
  const bool LIST = sparse.count != nullptr;
  /* bit 4 of useNext (dense results, no pre-fill: the host knows that this kernel takes the batch): a round's counts and
   * ranges are stored once, 64 consecutive k-mers -- whole lines -- per wave instruction, when the round's survivors are done
   * (a survivor leaves its final range in its slot), instead of a pre-filled array and a store at its k-mer number per hit:
   * every k-mer gets its range -- {1, 0} without hits -- and its count */
  const bool WHOLE = (useNext & 16u) != 0u && !LIST;
  if (threadIdx.x == 0) sWavesDone = 0u;
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  if (threadIdx.x < 17) sLevelAt[threadIdx.x] = threadIdx.x >= 1u ? awfmLengthTableAt(threadIdx.x) : 0ull;
  stageMaskTable(sMask);
  nucStageSuper<NARROW>(ix, sSuper);
  if (PAIR) pairStageTables<NARROW, 16u>(ix, sPairC, sPairSuper);
  __syncthreads();
  const unsigned DK = ix.deepK;
  const unsigned long long charsBytes = offsets[numQueries]; /* uniform */
  const unsigned lane = threadIdx.x & 63u, gl = threadIdx.x % G, firstSlice = gl;
  const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  unsigned hitFill = 0, keptHere = 0; /* wave-uniform */
  auto flushHits = [&]() {
    if (hitFill != 0u) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      unsigned listBase = 0;
      if (lane == 0) listBase = atomicAdd(sparse.count, hitFill);
      listBase = (unsigned)__builtin_amdgcn_readfirstlane((int)listBase);
      if (lane < hitFill && listBase + lane < sparse.cap) {
        sparse.kmers[listBase + lane] = sHitKmers[w][lane];
        sparse.ranges[listBase + lane] = make_ulonglong2(sHitRanges[w][lane][0], sHitRanges[w][lane][1]);
      }
      __builtin_amdgcn_wave_barrier();
      hitFill = 0;
    }
  };
  const unsigned long long waveStride = 1024ull * gridDim.x;
  for (unsigned long long tw = 1024ull * blockIdx.x + 256ull * w; tw < numQueries; tw += waveStride) {
    unsigned long long codes[4];
    unsigned len[4]; /* 0: not looked up (not in the batch, or the general kernel's) */
    bool general[4];
#pragma unroll
    for (unsigned i = 0; i < 4u; i++) {
      const unsigned long long q = tw + 64ull * i + lane;
      const bool inBatch = q < numQueries;
      unsigned long long start = 0, l = 0;
      if (inBatch) {
        start = offsets[q];
        l = offsets[q + 1ull] - start;
      }
      const unsigned n = l > 33ull ? 33u : (unsigned)l;
      const bool inRange = inBatch && n >= 1u && n <= 32u;
      unsigned bad = 0;
      codes[i] = 0;
      if (inRange) {
        if (start + 48ull <= charsBytes) decodeKmerWide(chars, start, n, codes[i], bad);
        else decodeKmer(chars, start, n, codes[i], bad);
      }
      general[i] = inBatch && (!inRange || bad != 0u);
      len[i] = inRange && bad == 0u ? n : 0u;
    }
    uint2 entry[4];
#pragma unroll
    for (unsigned i = 0; i < 4u; i++) entry[i] = *(len[i] != 0u ? mixedEntryAt(ix, lengthTable, sLevelAt, len[i], codes[i]) : lengthTable);
    /* the entries settle most k-mers: their hits are stored right away (dense: under their numbers; list: one reservation
     * per round for all of them).  Survivors go to the wave's slots, the rest of what is unsettled to the general kernel's list. */
    unsigned stotal = 0, htotal = 0;
    unsigned long long hmask[4];
    unsigned hbefore[4];
    /* what the entry settled: {first position, count} to report now (count 0: nothing), or -- WHOLE -- {slot, ~0} of a survivor */
    pos_t nowSp[4], nowLen[4];
#pragma unroll
    for (unsigned i = 0; i < 4u; i++) {
      const unsigned long long q = tw + 64ull * i + lane;
      const MixedVerdict<pos_t> v = mixedRead<pos_t>(ix, useNext, len[i], codes[i], entry[i]);
      nowSp[i] = v.sp;
      nowLen[i] = v.hitNow ? v.length : (pos_t)0;
      if (LIST) {
        hmask[i] = __ballot(v.hitNow);
        hbefore[i] = htotal;
        htotal += (unsigned)__popcll(hmask[i]);
      } else if (v.hitNow && !WHOLE) {
        if (ranges) ranges[q] = make_ulonglong2((unsigned long long)v.sp, (unsigned long long)v.sp + (unsigned long long)v.length - 1ull);
        if (counts) counts[q] = (unsigned)v.length;
      }
      const unsigned long long smask = __ballot(v.survives);
      const unsigned rank = stotal + (unsigned)__popcll(smask & ((1ull << lane) - 1ull));
      stotal += (unsigned)__popcll(smask);
      if (v.survives && rank < kMixedSlots) {
        sRem[w][rank] = codes[i] >> (2u * DK);
        sLeft[w][rank] = (unsigned char)(len[i] - DK);
        sNum[w][rank] = (unsigned)q;
        sSp[w][rank] = v.sp;
        sEp[w][rank] = v.sp + v.length - (pos_t)1;
      }
      if (WHOLE && v.survives && rank < kMixedSlots) {
        nowSp[i] = (pos_t)rank;
        nowLen[i] = ~(pos_t)0;
      }
      const bool left = general[i] || (v.survives && rank >= kMixedSlots);
      const unsigned long long lmask = __ballot(left);
      if (lmask != 0ull) { /* wave-uniform; rare */
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(leftoverCount, (unsigned)__popcll(lmask));
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        /* the list is filled from its END: searchKernel<INDIRECT> reads the last *count records of the array */
        if (left) leftover[numQueries - 1ull - (base + (unsigned)__popcll(lmask & ((1ull << lane) - 1ull)))] = q;
      }
    }
    if (LIST && htotal != 0u) { /* wave-uniform */
      unsigned listBase = 0;
      if (lane == 0) listBase = atomicAdd(sparse.count, htotal);
      listBase = (unsigned)__builtin_amdgcn_readfirstlane((int)listBase);
#pragma unroll
      for (unsigned i = 0; i < 4u; i++)
        if (nowLen[i] != 0) { /* (LIST: never a slot marker) */
          const unsigned at = listBase + hbefore[i] + (unsigned)__popcll(hmask[i] & ((1ull << lane) - 1ull));
          if (at < sparse.cap) {
            sparse.kmers[at] = (unsigned)(tw + 64ull * i + lane);
            sparse.ranges[at] = make_ulonglong2((unsigned long long)nowSp[i], (unsigned long long)nowSp[i] + (unsigned long long)nowLen[i] - 1ull);
          }
        }
    }
    const unsigned inRound = stotal < kMixedSlots ? stotal : kMixedSlots;
    keptHere += inRound;
    if (inRound != 0u) { /* wave-uniform */
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      /* (Round 5 measured what bounds these steps: with the survivors dropped (a diagnostic build, wrong results)
       * the kernel takes 2.8 of its 6.7 ms per 10^8 8..30-mers, so the steps of 3.3 * 10^7 survivors take 3.9: 1.2 * 10^8 block
       * lines at 3.1 * 10^7 a second.  TWO k-mers per group of 4 lanes, the loads of both requested before either is ranked --
       * 32 chains a wave, at 4 and at 5 waves per SIMD -- took 6.74 and 6.60 ms: the same.  The chains in flight are not the
       * bound.) */
      /* The survivors of a round have one to eight pair steps to go (and many end after the first): taken 16 at a time, a
       * pass lasts as long as its longest chain while the other groups of 4 lanes idle -- 10^8 8..30-mers, half of them
       * drawn from the text: 6.4 ms.  So a group that is done with its k-mer takes the next slot at once, and every
       * iteration of the loop is one memory round for 16 k-mers as long as there are 16.  A k-mer with an odd number of
       * characters to go is parked when one is left (its range back into its slot, the slot into sOdd) and takes that
       * step through the one-letter image after the loop: a wave iteration issues one kind of block read. */
      constexpr unsigned kGroups = 64u / G;
      const unsigned leader = lane & ~(unsigned)(G - 1);
      unsigned nextSlot = kGroups, oddCount = 0; /* wave-uniform */
      unsigned mySlot = lane / G;
      bool live = mySlot < inRound;
      pos_t sp = 1, ep = 0;
      unsigned long long rem = 0;
      unsigned index = 0;
      int pos = -1;
      auto take = [&]() {
        rem = sRem[w][mySlot];
        index = sNum[w][mySlot];
        sp = sSp[w][mySlot];
        ep = sEp[w][mySlot];
        pos = (int)sLeft[w][mySlot] - 1;
      };
      auto report = [&](const bool hit) { /* wave-uniform call; hit: in lane 0 of the groups that have one */
        if (LIST) {
          const unsigned long long hitMask = __ballot(hit);
          if (hitMask != 0ull) { /* wave-uniform; at most 16 hits */
            const unsigned hits = (unsigned)__builtin_amdgcn_readfirstlane((int)__popcll(hitMask));
            if (hit) {
              const unsigned at = hitFill + (unsigned)__popcll(hitMask & ((1ull << lane) - 1ull));
              sHitKmers[w][at] = index;
              sHitRanges[w][at][0] = (unsigned long long)sp;
              sHitRanges[w][at][1] = (unsigned long long)ep;
            }
            hitFill = (unsigned)__builtin_amdgcn_readfirstlane((int)(hitFill + hits));
          }
          if (hitFill + kGroups > kHitBuffer) flushHits();
        } else if (hit && !WHOLE) {
          if (ranges) ranges[index] = make_ulonglong2((unsigned long long)sp, (unsigned long long)ep);
          if (counts) counts[index] = (unsigned)(ep - sp + (pos_t)1);
        }
      };
      if (live) take();
      const int stepChars = PAIR ? 2 : 1;
      while (__ballot(live) != 0ull) { /* wave-uniform */
        if (live && pos >= stepChars - 1) { /* (a k-mer in a group is alive: sp <= ep) */
          if (PAIR) {
            const unsigned c2 = (unsigned)rem & 3u, c1 = (unsigned)(rem >> 2) & 3u;
            if (pairSearchStep<NARROW>(ix, sPairC, sPairSuper, sMask, gl, c1 * 4u + c2, sp, ep) == kPairFlagged) {
              nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, c2, sp, ep);
              if (sp <= ep) nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, c1, sp, ep);
            }
            pos -= 2;
            rem >>= 4;
          } else {
            nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, (unsigned)rem & 3u, sp, ep);
            pos--;
            rem >>= 2;
          }
        }
        /* what the step (or the slot as it was taken) leaves: the group is done with its k-mer when the range is empty, when
         * no character is left (a hit), or when one is left that the pair steps cannot take (parked) */
        const bool gone = live && (sp > ep || pos < stepChars - 1);
        const bool park = gone && sp <= ep && pos == 0; /* PAIR only */
        report(gone && gl == 0 && sp <= ep && pos < 0);
        if (WHOLE && gone && !park && gl == 0) { /* its final range (empty: it died on the way; else pos < 0) back into its slot */
          sSp[w][mySlot] = sp;
          sEp[w][mySlot] = ep;
        }
        const unsigned long long parkMask = __ballot(park && gl == 0);
        if (parkMask != 0ull) { /* wave-uniform */
          if (park && gl == 0) {
            sSp[w][mySlot] = sp;
            sEp[w][mySlot] = ep;
            sRem[w][mySlot] = rem;
            sOdd[w][oddCount + (unsigned)__popcll(parkMask & ((1ull << lane) - 1ull))] = (unsigned char)mySlot;
          }
          oddCount += (unsigned)__popcll(parkMask);
        }
        const unsigned long long goneMask = __ballot(gone && gl == 0);
        if (gone) {
          mySlot = nextSlot + (unsigned)__popcll(goneMask & ((1ull << leader) - 1ull));
          live = mySlot < inRound;
          sp = 1;
          ep = 0;
          pos = -1;
          if (live) take();
        }
        nextSlot += (unsigned)__popcll(goneMask);
      }
      if (oddCount != 0u) { /* wave-uniform: the parked k-mers' last step */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (unsigned pass = 0; pass < oddCount; pass += kGroups) { /* wave-uniform */
          const unsigned k = pass + lane / G;
          const bool parked = k < oddCount;
          sp = 1;
          ep = 0;
          if (parked) {
            mySlot = sOdd[w][k];
            take();
            nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, (unsigned)rem & 3u, sp, ep);
          }
          report(parked && gl == 0 && sp <= ep);
          if (WHOLE && parked && gl == 0) {
            sSp[w][mySlot] = sp;
            sEp[w][mySlot] = ep;
          }
        }
      }
      __builtin_amdgcn_wave_barrier(); /* the slots are written again by the next round */
    }
    if (WHOLE) { /* uniform */
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (unsigned i = 0; i < 4u; i++) {
        const unsigned long long q = tw + 64ull * i + lane;
        const bool slot = nowLen[i] == ~(pos_t)0;
        const unsigned at = slot ? (unsigned)nowSp[i] : 0u;
        const pos_t fsp = slot ? sSp[w][at] : nowSp[i], fep = slot ? sEp[w][at] : nowSp[i] + nowLen[i] - (pos_t)1;
        const bool hit = slot ? fsp <= fep : nowLen[i] != 0;
        if (q < numQueries) { /* (k-mers left to the general kernel: no hit until it stores what it finds) */
          if (ranges) ranges[q] = hit ? make_ulonglong2((unsigned long long)fsp, (unsigned long long)fep) : make_ulonglong2(1ull, 0ull);
          if (counts) counts[q] = hit ? (unsigned)(fep - fsp + (pos_t)1) : 0u;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (lane == 0 && keptHere) atomicAdd(&keptCounters[((blockIdx.x * 4u + w) % kFusedCounters) * 16u], keptHere);
  if (LIST) { /* the waves' leftovers in one reservation, as in orderedSearchKernel */
    if (lane == 0) sHitLeft[w] = hitFill;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    unsigned arrived = 0;
    if (lane == 0) arrived = atomicAdd(&sWavesDone, 1u);
    arrived = (unsigned)__builtin_amdgcn_readfirstlane((int)arrived);
    if (arrived == 3u) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      unsigned total = 0;
      for (unsigned v = 0; v < 4u; v++) total += sHitLeft[v];
      if (total != 0u) {
        unsigned listBase = 0;
        if (lane == 0) listBase = atomicAdd(sparse.count, total);
        listBase = (unsigned)__builtin_amdgcn_readfirstlane((int)listBase);
        unsigned before = 0;
        for (unsigned v = 0; v < 4u; v++) {
          const unsigned n = sHitLeft[v];
          if (lane < n && listBase + before + lane < sparse.cap) {
            sparse.kmers[listBase + before + lane] = sHitKmers[v][lane];
            sparse.ranges[listBase + before + lane] = make_ulonglong2(sHitRanges[v][lane][0], sHitRanges[v][lane][1]);
          }
          before += n;
        }
      }
    }
  }
}

/* how many of `samples` k-mers at a fixed stride are not settled by their table entry (survivors, and the general kernel's):
 * says beforehand whether the batch is one for mixedLookupSearchKernel */
__global__ void __launch_bounds__(256)
    mixedSampleAliveKernel(const DevIndex ix, const uint2 *__restrict__ lengthTable, const unsigned char *__restrict__ chars,
                           const unsigned long long *__restrict__ offsets, const unsigned long long numQueries, const unsigned useNext,
                           const unsigned samples, unsigned long long *__restrict__ aliveOut /* zeroed; the low half is the count */,
                           unsigned long long *__restrict__ verdictHost, const unsigned searchNumber) {
  __shared__ unsigned long long sLevelAt[17];
  __shared__ unsigned sAlive;
  if (threadIdx.x < 17) sLevelAt[threadIdx.x] = threadIdx.x >= 1u ? awfmLengthTableAt(threadIdx.x) : 0ull;
  if (threadIdx.x == 0) sAlive = 0u;
  __syncthreads();
  const unsigned j = blockIdx.x * 256u + threadIdx.x;
  bool alive = false;
  if (j < samples) {
    const unsigned long long q = (unsigned long long)j * (numQueries / samples);
    const unsigned long long start = offsets[q], l = offsets[q + 1ull] - start;
    if (l >= 1ull && l <= 32ull) {
      unsigned long long codes;
      unsigned bad;
      decodeKmer(chars, start, (unsigned)l, codes, bad);
      alive = bad != 0u || mixedRead<unsigned long long>(ix, useNext, (unsigned)l, codes, *mixedEntryAt(ix, lengthTable, sLevelAt, (unsigned)l, codes)).survives;
    } else {
      alive = true;
    }
  }
  const unsigned n = (unsigned)__popcll(__ballot(alive));
  if ((threadIdx.x & 63u) == 0 && n) atomicAdd(&sAlive, n);
  __syncthreads();
  if (threadIdx.x == 0) { /* {1, count} in one atomic: the workgroup that sees all the others in publishes the verdict (lookupPrepKernel) */
    const unsigned long long old = atomicAdd(aliveOut, (1ull << 32) | (unsigned long long)sAlive);
    if ((unsigned)(old >> 32) == gridDim.x - 1u && verdictHost)
      *(volatile unsigned long long *)verdictHost = ((unsigned long long)searchNumber << 32) | (unsigned long long)((unsigned)old + sAlive);
  }
}

/* ---- what mixedLookupSearchKernel has to read (awfmGpuMixedLookupLineTally) ----
 * The same reading of every k-mer -- decode, table entry, verdict, the survivors' steps -- one group of 4 lanes per k-mer,
 * with every 128-B line marked in a bitmap: the lines of the length tables, of the deeper table, and per search level
 * (characters taken since the table) the lines of the pair image and of the one-letter image.  Not for timing. */
struct MixedTouch {
  unsigned long long *lengthLines; /* bit per 128-B line of the length tables */
  unsigned long long *deepLines;   /* ... of the deeper table */
  unsigned long long *pairLines;   /* [level][pairWords]: bit per pair block (one line each) */
  unsigned long long *nucLines;    /* [level][nucWords]: bit per line of the one-letter image (two 64-B blocks) */
  unsigned long long pairWords, nucWords;
  unsigned long long *sums; /* [0] k-mers still alive after their entry, [1] k-mers with hits, [2] k-mers left to the general kernel,
                             * [7] block lines the survivors' steps read (one per step, two when sp - 1 and ep lie in different blocks) */
};
constexpr unsigned kMixedTouchLevels = 17; /* characters a k-mer of up to 32 can have to go beyond a table */

template <bool NARROW>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80)))
    mixedLookupTallyKernel(const DevIndex ix, const uint2 *__restrict__ lengthTable, const unsigned char *__restrict__ chars,
                           const unsigned long long *__restrict__ offsets, const unsigned long long numQueries, const unsigned useNext,
                           const MixedTouch touch) {
  constexpr int G = 4;
  typedef typename PositionType<NARROW>::type pos_t;
  __shared__ unsigned long long sC[24];
  __shared__ unsigned sMask[(kBlockMask + 1) * kSlices];
  __shared__ unsigned long long sSuper[!NARROW ? kMaxNucSuper * 4 : 1];
  __shared__ unsigned long long sPairC[16];
  extern __shared__ unsigned sPairSuper[];
  __shared__ unsigned long long sLevelAt[17];
  const bool PAIR = ix.pairBlocks != nullptr && (useNext & 2u) == 0u;
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  if (threadIdx.x < 17) sLevelAt[threadIdx.x] = threadIdx.x >= 1u ? awfmLengthTableAt(threadIdx.x) : 0ull;
  stageMaskTable(sMask);
  nucStageSuper<NARROW>(ix, sSuper);
  if (PAIR) pairStageTables<NARROW, 16u>(ix, sPairC, sPairSuper);
  __syncthreads();
  const unsigned DK = ix.deepK;
  const unsigned gl = threadIdx.x % G;
  auto mark = [&](unsigned long long *bits, unsigned long long line) {
    if (gl == 0) atomicOr(bits + (line >> 6), 1ull << (line & 63ull));
  };
  unsigned alive = 0, hits = 0, general = 0, reads = 0; /* counted in lane 0 of a group */
  const unsigned long long groups = (unsigned long long)gridDim.x * (256u / G);
  for (unsigned long long q = ((unsigned long long)blockIdx.x * 256u + threadIdx.x) / G; q < numQueries; q += groups) {
    const unsigned long long start = offsets[q], l = offsets[q + 1ull] - start;
    unsigned long long codes = 0;
    unsigned bad = 0;
    const bool inRange = l >= 1ull && l <= 32ull;
    if (inRange) decodeKmer(chars, start, (unsigned)l, codes, bad);
    if (!inRange || bad != 0u) {
      general += gl == 0 ? 1u : 0u;
      continue;
    }
    const unsigned len = (unsigned)l;
    const uint2 *at = mixedEntryAt(ix, lengthTable, sLevelAt, len, codes);
    if (len >= DK) mark(touch.deepLines, (unsigned long long)(at - (const uint2 *)ix.deepSeed) >> 4);
    else mark(touch.lengthLines, (unsigned long long)(at - lengthTable) >> 4);
    const uint2 entry = *at;
    const MixedVerdict<pos_t> v = mixedRead<pos_t>(ix, useNext, len, codes, entry);
    if (v.hitNow) hits += gl == 0 ? 1u : 0u;
    if (!v.survives) continue;
    alive += gl == 0 ? 1u : 0u;
    pos_t sp = v.sp, ep = v.sp + v.length - (pos_t)1;
    unsigned long long rem = codes >> (2u * DK);
    const int first = (int)(len - DK) - 1;
    int pos = first;
    auto touchPair = [&](unsigned level) {
      unsigned long long *bits = touch.pairLines + (level < kMixedTouchLevels ? level : kMixedTouchLevels - 1u) * touch.pairWords;
      mark(bits, (unsigned long long)(sp - (pos_t)1) >> kBlockShift);
      mark(bits, (unsigned long long)ep >> kBlockShift);
      reads += gl == 0 ? (((sp - (pos_t)1) >> kBlockShift) != (ep >> kBlockShift) ? 2u : 1u) : 0u;
    };
    auto touchNuc = [&](unsigned level) {
      unsigned long long *bits = touch.nucLines + (level < kMixedTouchLevels ? level : kMixedTouchLevels - 1u) * touch.nucWords;
      mark(bits, ((unsigned long long)(sp - (pos_t)1) >> kBlockShift) >> 1);
      mark(bits, ((unsigned long long)ep >> kBlockShift) >> 1);
      reads += gl == 0 ? (((sp - (pos_t)1) >> kBlockShift) != (ep >> kBlockShift) ? 2u : 1u) : 0u;
    };
    if (PAIR) {
      while (pos >= 1 && sp <= ep) {
        const unsigned c2 = (unsigned)rem & 3u, c1 = (unsigned)(rem >> 2) & 3u;
        touchPair((unsigned)(first - pos));
        if (pairSearchStep<NARROW>(ix, sPairC, sPairSuper, sMask, gl, c1 * 4u + c2, sp, ep) == kPairFlagged) {
          touchNuc((unsigned)(first - pos));
          nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, gl, c2, sp, ep);
          if (sp <= ep) {
            touchNuc((unsigned)(first - pos) + 1u);
            nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, gl, c1, sp, ep);
          }
        }
        pos -= 2;
        rem >>= 4;
      }
      if (pos == 0 && sp <= ep) {
        touchNuc((unsigned)first);
        nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, gl, (unsigned)rem & 3u, sp, ep);
        pos--;
      }
    } else {
      while (pos >= 0 && sp <= ep) {
        touchNuc((unsigned)(first - pos));
        nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, gl, (unsigned)rem & 3u, sp, ep);
        pos--;
        rem >>= 2;
      }
    }
    if (sp <= ep) hits += gl == 0 ? 1u : 0u;
  }
  /* (the loop's `continue`s are per group: all lanes are here again) */
  unsigned long long a = alive, h = hits, n = general, r = reads;
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off);
    h += __shfl_down(h, off);
    n += __shfl_down(n, off);
    r += __shfl_down(r, off);
  }
  if ((threadIdx.x & 63u) == 0u) {
    if (a) atomicAdd(touch.sums, a);
    if (h) atomicAdd(touch.sums + 1, h);
    if (n) atomicAdd(touch.sums + 2, n);
    if (r) atomicAdd(touch.sums + 7, r);
  }
}

}  // namespace

#endif
