/*
 * awfm_amino_lookup_kernel.h -- "lookup first" for the amino alphabet (hits-only searches of large fixed-length batches).
 *
 * With the device-only deeper table (awfm_device.h: depth 7 for a Swiss-Prot-sized image, 20^7 entries of 8 bytes) 85 % of
 * random 10-mers end at their entry: no such 7-mer in the text.  The general kernel gets there one k-mer per lane group
 * and wave round -- a chain of characters, entry, steps with 256 lookups in flight per SIMD -- and takes 2.7 ms per 5 * 10^7
 * where the table gather itself is 1.3 ms.  This kernel is the amino twin of lookupSearchKernel (awfm_ordered_kernel.h): a
 * thread decodes FOUR consecutive k-mers from 16-byte loads, looks their four entries up at once, and the wave takes the
 * survivors (14 %: 36 of a round of 256) through their remaining steps out of LDS, 16 at a time (4 lanes per k-mer,
 * aminoStepAny: the step of the general kernel).  What it does not cover -- a character that is not one of the 20
 * letters among the table's characters, survivors beyond the 64 slots of a round -- goes to a list that the general kernel
 * searches afterwards (INDIRECT).  A sample decides, on the device, between this kernel and the general one for the
 * whole batch (lookupChosen).
 *
 * Results: every k-mer with hits gets the reference's range (same table entry as the index's own table + steps, same
 * steps behind it: ref src/AwFmKmerTable.c:37-51, src/AwFmSearch.c:105-159, src/AwFmParallelSearch.c:273-313); a k-mer
 * without hits has count 0 and an empty range (the hits-only contract of awfmGpuSearchHits).
 */
#ifndef AWFM_AMINO_LOOKUP_KERNEL_H
#define AWFM_AMINO_LOOKUP_KERNEL_H

#include "awfm_ordered_kernel.h"

namespace {

/* survivors a wave takes through the steps per round: as many as the round has k-mers (round 5; 64 until then, the rest went
 * to the general kernel's list -- fine for a Swiss-Prot-sized text, where 15 % of random 10-mers survive their entry, not for
 * 2 * 10^9 residues, where 79 % do) */
constexpr unsigned kAminoSlots = 256;

/* letter indices (ref src/AwFmLetter.c:55-67) of the K characters of k-mer i of a thread's four: idx = the table index over
 * the last DK of them (leftmost first), lead = the K - DK before them, 5 bits each, the one stepped first in bits 4..0;
 * bad: one of the table's characters is not one of the 20 letters */
template <unsigned K, class Bytes>
__device__ __forceinline__ void aminoDecode(const AminoShared &t, const Bytes &byteAt, unsigned kmer, unsigned DK, unsigned &idx,
                                            unsigned long long &lead, bool &bad) {
  idx = 0;
  lead = 0;
  bad = false;
#pragma unroll
  for (unsigned j = 0; j < K; j++) {
    const unsigned c = byteAt(kmer * K + j);
    const unsigned letter = aminoLetterIndex(t, c);
    const bool inTable = j + DK >= K; /* uniform */
    if (inTable) {
      idx = idx * 20u + letter;
      bad |= letter >= 20u;
    } else {
      lead = (lead << 5) | letter; /* character 0 ends up highest: the last one stepped */
    }
  }
}

/* NARROW = false (round 6): images of 2^32 positions and more -- 64-bit ranges in the slots and the steps, the table's entries in
 * their packed form (awfm_device.h: aminoWidePack); 4 waves per SIMD (the LDS of 5 workgroups would fit a CU, the registers of 5 waves do not) */
template <unsigned K, bool NARROW = true>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(NARROW ? 6 : 4, 8))) /* 6 workgroups' LDS fit a CU: 6 waves per SIMD, 80 registers */
    aminoLookupSearchKernel(const DevIndex ix, const unsigned char *__restrict__ chars, const unsigned long long numQueries,
                            const unsigned *__restrict__ sampleAlive, const unsigned samples, ulonglong2 *__restrict__ ranges,
                            unsigned *__restrict__ counts, const SparseOut sparse, unsigned long long *__restrict__ leftover,
                            unsigned *__restrict__ leftoverCount, unsigned *__restrict__ keptCounters) {
  constexpr int G = 4;
  typedef typename PositionType<NARROW>::type pos_t;
  __shared__ unsigned long long sC[24];
  __shared__ AminoShared sAmino;
  __shared__ unsigned sMask[(kBlockMask + 1) * kSlices];
  __shared__ unsigned long long sLead[4][kAminoSlots];
  __shared__ unsigned sNum[4][kAminoSlots];
  __shared__ pos_t sSp[4][kAminoSlots], sEp[4][kAminoSlots];
  constexpr unsigned kHitBuffer = 32;
  __shared__ unsigned sHitKmers[4][kHitBuffer];
  __shared__ unsigned long long sHitRanges[4][kHitBuffer][2];
  __shared__ unsigned sHitLeft[4], sWavesDone;
  if (!lookupChosen(sampleAlive, samples, true)) return; /* this batch is the general kernel's (uniform) */
  const bool LIST = sparse.count != nullptr;
  if (threadIdx.x == 0) sWavesDone = 0u;
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  aminoStageTables(sAmino);
  stageMaskTable(sMask);
  __syncthreads();
  const unsigned DK = ix.deepK;
  constexpr unsigned kLoads = (K + 1u + 3u) / 4u;
  const unsigned shift = (unsigned)((unsigned long long)chars & 3ull);
  typedef const Dwords4 __attribute__((address_space(1))) *GlobalDwords4;
  const unsigned lane = threadIdx.x & 63u, gl = threadIdx.x % G;
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
  const unsigned long long waveStride = 4ull * gridDim.x * 256ull;
  for (unsigned long long tw = 4ull * ((unsigned long long)blockIdx.x * 256ull + (threadIdx.x & ~63u)); tw < numQueries; tw += waveStride) {
    const unsigned long long t = tw + 4ull * lane;
    unsigned idx[4];
    unsigned long long lead[4];
    bool bad[4];
    if (t * K + 16ull * kLoads <= numQueries * K) { /* the 16-byte loads from the aligned-down start stay inside the batch */
      const GlobalDwords4 from = (GlobalDwords4)(((unsigned long long)chars + t * K) & ~3ull);
      unsigned dw[kLoads * 4u + 1u];
#pragma unroll
      for (unsigned j = 0; j < kLoads; j++) {
        const Dwords4 q = from[j];
        dw[4u * j] = q.x;
        dw[4u * j + 1u] = q.y;
        dw[4u * j + 2u] = q.z;
        dw[4u * j + 3u] = q.w;
      }
      dw[kLoads * 4u] = 0u;
      unsigned al[K + 1u]; /* the 4 K bytes at dword alignment */
#pragma unroll
      for (unsigned j = 0; j < K; j++) al[j] = __builtin_amdgcn_alignbyte(dw[j + 1u], dw[j], shift);
      al[K] = 0u;
      auto byteAt = [&](unsigned b) -> unsigned { return (al[b >> 2] >> (8u * (b & 3u))) & 0xFFu; }; /* compile-time b after unrolling */
#pragma unroll
      for (unsigned i = 0; i < 4u; i++) aminoDecode<K>(sAmino, byteAt, i, DK, idx[i], lead[i], bad[i]);
    } else {
#pragma unroll
      for (unsigned i = 0; i < 4u; i++) {
        idx[i] = 0;
        lead[i] = 0;
        bad[i] = false;
        if (t + i < numQueries) {
          const unsigned char *at = chars + (t + i) * K;
          auto byteAt = [&](unsigned b) -> unsigned { return at[b]; };
          aminoDecode<K>(sAmino, byteAt, 0u, DK, idx[i], lead[i], bad[i]);
        }
      }
    }
    uint2 entry[4];
#pragma unroll
    for (unsigned i = 0; i < 4u; i++) entry[i] = ((const uint2 *)ix.deepSeed)[t + i < numQueries && !bad[i] ? idx[i] : 0u];
    unsigned stotal = 0;
#pragma unroll
    for (unsigned i = 0; i < 4u; i++) {
      const bool inBatch = t + i < numQueries;
      /* alive after the entry: its range holds something and -- tables with next-step bits -- still will after the next
       * letter (the first of the lead's: bits 4..0; a k-mer that ends at the table has no next letter) */
      const unsigned long long length = aminoDeepLength(ix, entry[i]);
      const bool survives = inBatch && !bad[i] && length != 0ull && (K == DK || aminoDeepNextBit(ix, entry[i], (unsigned)lead[i] & 31u));
      const unsigned long long smask = __ballot(survives);
      const unsigned rank = stotal + (unsigned)__popcll(smask & ((1ull << lane) - 1ull));
      stotal += (unsigned)__popcll(smask);
      if (survives && rank < kAminoSlots) {
        sLead[w][rank] = lead[i];
        sNum[w][rank] = (unsigned)(t + i);
        const unsigned long long first = aminoDeepSp(ix, entry[i]);
        sSp[w][rank] = (pos_t)first;
        sEp[w][rank] = (pos_t)(first + length - 1ull);
      }
      /* the general kernel's: a character among the table's that is not one of the 20 letters; a survivor without a slot */
      const bool left = inBatch && (bad[i] || (survives && rank >= kAminoSlots));
      const unsigned long long lmask = __ballot(left);
      if (lmask != 0ull) { /* wave-uniform; rare */
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(leftoverCount, (unsigned)__popcll(lmask));
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        /* the list is filled from its END: searchKernel<INDIRECT> reads the last *count records of the array */
        if (left) leftover[numQueries - 1ull - (base + (unsigned)__popcll(lmask & ((1ull << lane) - 1ull)))] = (unsigned long long)(t + i);
      }
    }
    const unsigned inRound = stotal < kAminoSlots ? stotal : kAminoSlots;
    keptHere += inRound;
    if (inRound != 0u) { /* wave-uniform */
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      /* 16 groups of 4 lanes, each with one survivor; a group that is done with its k-mer -- the range is empty (most are
       * after one step) or no character is left -- takes the next slot at once (as mixedLookupSearchKernel's groups do), so
       * that every iteration of the loop is one memory round for 16 k-mers as long as there are 16 */
      constexpr unsigned kGroups = 64u / G;
      const unsigned leader = lane & ~(unsigned)(G - 1);
      unsigned nextSlot = kGroups; /* wave-uniform */
      unsigned mySlot = lane / G;
      bool live = mySlot < inRound;
      pos_t sp = 1, ep = 0;
      unsigned long long rem = 0;
      unsigned index = 0;
      int pos = -1;
      auto take = [&]() {
        rem = sLead[w][mySlot];
        index = sNum[w][mySlot];
        sp = sSp[w][mySlot];
        ep = sEp[w][mySlot];
        pos = (int)(K - DK) - 1;
      };
      if (live) take();
      while (__ballot(live) != 0ull) { /* wave-uniform */
        if (live && pos >= 0) { /* (a k-mer in a group is alive: sp <= ep) */
          aminoStepAny<G, NARROW>(ix, sC, sAmino, sMask, gl, (unsigned)rem & 31u, sp, ep);
          pos--;
          rem >>= 5;
        }
        const bool gone = live && (sp > ep || pos < 0);
        const bool hit = gone && gl == 0 && sp <= ep;
        if (LIST) {
          const unsigned long long hitMask = __ballot(hit);
          if (hitMask != 0ull) { /* wave-uniform; at most 16 hits an iteration */
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
        } else if (hit) {
          if (ranges) ranges[index] = make_ulonglong2((unsigned long long)sp, (unsigned long long)ep);
          if (counts) counts[index] = (unsigned)(ep - sp + (pos_t)1);
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
      __builtin_amdgcn_wave_barrier(); /* the slots are written again by the next round */
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

}  // namespace

#endif
