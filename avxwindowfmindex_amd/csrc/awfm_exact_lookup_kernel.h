/*
 * awfm_exact_lookup_kernel.h -- "lookup first" for awfmGpuSearch: the EXACT final range of every k-mer through the tables.
 *
 * awfmGpuSearch owes every k-mer the range the reference's stepping ends in -- for a k-mer without hits the FIRST empty range
 * (ref src/AwFmParallelSearch.c:273-313 keeps it; src/AwFmSearch.c:317-358 returns it) --, which the hits-only kernels
 * (lookupSearchKernel, mixedLookupSearchKernel) do not produce: they drop a k-mer on a clear next-step bit and let a pair step
 * end in the pair's empty range.  The tables themselves are exact: an entry is {sp, length} of what the reference reaches with
 * its stop-at-the-first-empty-range rule, and an empty range of that stepping is always {sp, sp - 1} (awfm_device.h), so a
 * length of 0 IS the reference's final range.  This kernel is mixedLookupSearchKernel without the shortcuts:
 *   one entry per k-mer -- the deeper table's over its last deepK characters, or, for a k-mer shorter than that, the entry of
 *   its own length's table (awfmGpuBuildLengthTables) --; an empty entry, or one that covers the whole k-mer, is the result;
 *   the others (10^8 random 21-mers against 3.1 Gbp: 51 %) go on out of LDS, 16 at a time, each taking the next one's place as
 *   soon as it is done: EXACT pair steps (pairSearchStep<true, true>: a k-mer that dies inside a pair gets the range of the
 *   single step that emptied it), flagged blocks and the odd last character through the one-letter image;
 *   every k-mer's {sp, ep} and count stored under its number (batch order: whole lines).
 * K-mers with a character that is not a,c,g,t,u, none or more than 32 characters go to a list for the general kernel.
 * Fixed-length batches (offsets == NULL) and CSR ones; images with an 8-byte deeper table (DevIndex::deepNarrow 1 or 2),
 * 32- and 64-bit positions (round 6).
 */
#ifndef AWFM_EXACT_LOOKUP_KERNEL_H
#define AWFM_EXACT_LOOKUP_KERNEL_H

#include "awfm_mixed_lookup_kernel.h"

namespace {

template <bool NARROW>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80)))
    exactLookupSearchKernel(const DevIndex ix, const uint2 *__restrict__ lengthTable, const unsigned char *__restrict__ chars,
                            const unsigned long long *__restrict__ offsets, const unsigned fixedLength,
                            const unsigned long long numQueries, const unsigned pairOff, ulonglong2 *__restrict__ ranges,
                            unsigned *__restrict__ counts, unsigned long long *__restrict__ leftover,
                            unsigned *__restrict__ leftoverCount) {
  constexpr int G = 4;
  /* NARROW: 32-bit positions (awfmImageNarrow); otherwise (round 6) the 64-bit arithmetic of ref src/AwFmIndex.h:88-91; the
   * entries are read in the format the image built them (DevIndex::deepNarrow) either way */
  typedef typename PositionType<NARROW>::type pos_t;
  __shared__ unsigned long long sC[24];
  __shared__ unsigned sMask[(kBlockMask + 1) * kSlices];
  __shared__ unsigned long long sSuper[NARROW ? 1 : kMaxNucSuper * 4];
  __shared__ unsigned long long sPairC[16];
  extern __shared__ unsigned sPairSuper[];
  __shared__ unsigned long long sLevelAt[17];
  __shared__ unsigned long long sRem[4][kMixedSlots];
  __shared__ pos_t sSp[4][kMixedSlots], sEp[4][kMixedSlots];
  __shared__ unsigned char sLeft[4][kMixedSlots], sOdd[4][kMixedSlots];
  const bool PAIR = ix.pairBlocks != nullptr && pairOff == 0u;
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  if (threadIdx.x < 17) sLevelAt[threadIdx.x] = threadIdx.x >= 1u ? awfmLengthTableAt(threadIdx.x) : 0ull;
  stageMaskTable(sMask);
  nucStageSuper<NARROW>(ix, sSuper);
  if (PAIR) pairStageTables<NARROW, 16u>(ix, sPairC, sPairSuper);
  __syncthreads();
  const unsigned DK = ix.deepK;
  const unsigned long long charsBytes = offsets ? offsets[numQueries] : numQueries * (unsigned long long)fixedLength; /* uniform */
  const unsigned lane = threadIdx.x & 63u, gl = threadIdx.x % G, firstSlice = gl;
  const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  auto store = [&](const unsigned long long q, const pos_t sp, const pos_t ep) {
    if (ranges) ranges[q] = make_ulonglong2((unsigned long long)sp, (unsigned long long)ep);
    if (counts) counts[q] = sp <= ep ? (unsigned)(ep - sp + (pos_t)1) : 0u; /* ref src/AwFmIndexStruct.c:126-130 */
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
        if (offsets) {
          start = offsets[q];
          l = offsets[q + 1ull] - start;
        } else {
          start = q * (unsigned long long)fixedLength;
          l = fixedLength;
        }
      }
      const unsigned n = l > 33ull ? 33u : (unsigned)l;
      /* (a k-mer shorter than the deeper table needs the table of its own length) */
      const bool inRange = inBatch && n >= 1u && n <= 32u && (n >= DK || lengthTable != nullptr);
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
    for (unsigned i = 0; i < 4u; i++)
      entry[i] = *(len[i] != 0u ? mixedEntryAt(ix, lengthTable, sLevelAt, len[i], codes[i]) : (const uint2 *)ix.deepSeed);
    unsigned stotal = 0;
    /* a k-mer that ended at its entry: {sp, length} ({1, 0}: not looked up -- not in the batch, or the general kernel's, which
     * stores its own range later); a survivor: {slot, ~0} */
    pos_t nowSp[4], nowLen[4];
#pragma unroll
    for (unsigned i = 0; i < 4u; i++) {
      const unsigned long long q = tw + 64ull * i + lane;
      /* (the next-step bits are no use to an exact search: useNext 0) */
      const MixedVerdict<pos_t> v = mixedRead<pos_t>(ix, 0u, len[i], codes[i], entry[i]);
      const bool looked = len[i] != 0u;
      const bool survives = looked && len[i] > DK && v.length != 0;
      /* (a k-mer that ends at its entry: the entry is its final range.  Nothing is stored yet: a round's results go out
       * together, in whole lines, when its survivors are done -- the 10^8 k-mers of a batch stored one by one as they
       * finished were 2.7 x their bytes in partial-line writes) */
      const unsigned long long smask = __ballot(survives);
      const unsigned rank = stotal + (unsigned)__popcll(smask & ((1ull << lane) - 1ull));
      stotal += (unsigned)__popcll(smask);
      if (survives) { /* (a slot for every k-mer of the round) */
        sRem[w][rank] = codes[i] >> (2u * DK);
        sLeft[w][rank] = (unsigned char)(len[i] - DK);
        sSp[w][rank] = v.sp;
        sEp[w][rank] = v.sp + v.length - (pos_t)1;
      }
      nowSp[i] = survives ? (pos_t)rank : looked ? v.sp : (pos_t)1;
      nowLen[i] = survives ? ~(pos_t)0 : looked ? v.length : (pos_t)0;
      const unsigned long long lmask = __ballot(general[i]);
      if (lmask != 0ull) { /* wave-uniform; rare */
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(leftoverCount, (unsigned)__popcll(lmask));
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        /* the list is filled from its END: searchKernel<INDIRECT> reads the last *count records of the array */
        if (general[i]) leftover[numQueries - 1ull - (base + (unsigned)__popcll(lmask & ((1ull << lane) - 1ull)))] = q;
      }
    }
    if (stotal != 0u) { /* wave-uniform */
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      constexpr unsigned kGroups = 64u / G;
      const unsigned leader = lane & ~(unsigned)(G - 1);
      unsigned nextSlot = kGroups, oddCount = 0; /* wave-uniform */
      unsigned mySlot = lane / G;
      bool live = mySlot < stotal;
      pos_t sp = 1, ep = 0;
      unsigned long long rem = 0;
      int pos = -1;
      auto take = [&]() {
        rem = sRem[w][mySlot];
        sp = sSp[w][mySlot];
        ep = sEp[w][mySlot];
        pos = (int)sLeft[w][mySlot] - 1;
      };
      if (live) take();
      const int stepChars = PAIR ? 2 : 1;
      while (__ballot(live) != 0ull) { /* wave-uniform */
        if (live && pos >= stepChars - 1) { /* (a k-mer in a group is alive: sp <= ep) */
          if (PAIR) {
            const unsigned c2 = (unsigned)rem & 3u, c1 = (unsigned)(rem >> 2) & 3u;
            /* exact: a k-mer that dies inside the pair ends in the range the letter-by-letter stepping ends in */
            const PairStep did = pairSearchStep<NARROW, true>(ix, sPairC, sPairSuper, sMask, gl, c1 * 4u + c2, sp, ep, sC);
            if (did == kPairFlagged) nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, c2, sp, ep);
            if (did != kPairStepped && sp <= ep) nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, c1, sp, ep);
            pos -= 2;
            rem >>= 4;
          } else {
            nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, (unsigned)rem & 3u, sp, ep);
            pos--;
            rem >>= 2;
          }
        }
        /* done: the range is empty (the reference's final range), or no character is left; parked: one is left that the
         * pair steps cannot take (taken through the one-letter image after the loop: a wave iteration issues one kind of read) */
        const bool gone = live && (sp > ep || pos < stepChars - 1);
        const bool park = gone && sp <= ep && pos == 0; /* PAIR only */
        if (gone && gl == 0) { /* its final range, or where it is parked: back into its slot */
          sSp[w][mySlot] = sp;
          sEp[w][mySlot] = ep;
        }
        const unsigned long long parkMask = __ballot(park && gl == 0);
        if (parkMask != 0ull) { /* wave-uniform */
          if (park && gl == 0) {
            sRem[w][mySlot] = rem;
            sOdd[w][oddCount + (unsigned)__popcll(parkMask & ((1ull << lane) - 1ull))] = (unsigned char)mySlot;
          }
          oddCount += (unsigned)__popcll(parkMask);
        }
        const unsigned long long goneMask = __ballot(gone && gl == 0);
        if (gone) {
          mySlot = nextSlot + (unsigned)__popcll(goneMask & ((1ull << leader) - 1ull));
          live = mySlot < stotal;
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
            if (gl == 0) {
              sSp[w][mySlot] = sp;
              sEp[w][mySlot] = ep;
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (unsigned i = 0; i < 4u; i++) { /* the round's results: 64 consecutive k-mers per wave instruction */
      const unsigned long long q = tw + 64ull * i + lane;
      const bool slot = nowLen[i] == ~(pos_t)0;
      const unsigned at = slot ? (unsigned)nowSp[i] : 0u;
      const pos_t sp = slot ? sSp[w][at] : nowSp[i];
      const pos_t ep = slot ? sEp[w][at] : nowSp[i] + nowLen[i] - (pos_t)1;
      if (q < numQueries) store(q, sp, ep);
    }
    __builtin_amdgcn_wave_barrier(); /* the slots are written again by the next round */
  }
}

}  // namespace

#endif
