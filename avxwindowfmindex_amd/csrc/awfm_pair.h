/*
 * awfm_pair.h -- the pair image: two backward-search steps (or two LF steps) per block read.
 *
 * The backward search and the LF walk are chains of dependent block reads, one per character, and both run at the
 * rate the memory system serves those reads.  Prepending the characters c2 then c1 to a pattern P
 * maps the rows p of P's range with L[p] = c2 and L[LF(p)] = c1 -- the rows whose two preceding text characters
 * are c1 c2 -- order-preservingly onto the range of c1c2P.  So with the pair code of every BWT position,
 *     pair(p) = 4 * L[LF(p)] + L[p]      (letter indices a0 c1 g2 t3),
 * two steps of ref src/AwFmSearch.c:42-103 (two LF steps of :369-427) become one rank over a 16-letter sequence:
 *     sp'' = C2[c1c2] + rank_pair(sp - 1),   ep'' = C2[c1c2] + rank_pair(ep) - 1,   LF(LF(p)) = C2[pair(p)] + rank_pair(p) - 1
 * where C2[c1c2] = C[c1] + Occ(c1, C[c2]) is the first row of the suffixes that start with c1c2.  The results are
 * those of the two single steps, bit for bit; only the bytes read differ.
 *
 * Pair block = 128 B = one line per 128 BWT positions; slice k (positions 32k..32k+31) has two 16-B pieces, the plane
 * piece in the first 64-B sector of the line and the count piece in the second (pairPlanesAt / pairCountsAt):
 *   {b0, b1, b2, b3}   bit j of b_i = bit i of pair(128 blk + 32 k + j)
 *   {l, w0, w1, w2}    l = 24-bit count of letter k (a,c,g,t) before the block: in a block without a flag the low two
 *                      bits of a pair code are the position's own letter, so the block also gives the single step
 *                      LF(p) -- which the walk needs, because it must stop at a sampled position in between two steps;
 *                      w0..w2 = four 24-bit counts of the pairs 4k..4k+3 (c1 = k, c2 = 0..3) before the block
 *                      (bits 24 i .. 24 i + 23); all counts relative to the block's superblock of 2^24 positions (a
 *                      count before a block is at most 2^24 - 128: the superblock table of a GRCh38-sized image is
 *                      12 KB of LDS per workgroup instead of the 24 KB of 2^23-position superblocks);
 *                      bit 31 of l is set in every slice of a block that holds a position whose pair is not two of
 *                      a,c,g,t (ambiguity letter or sentinel at the position or at its LF image): such blocks are
 *                      stepped through the one-letter image, a letter at a time.
 * pairSuper[20 sb + i] = absolute count at the start of superblock sb of pair i (i < 16) or letter i - 16 (64-bit; a
 * 32-bit copy for LDS).
 *
 * A pattern that dies inside a pair step ends, in the reference's letter-by-letter stepping, in the empty range of the
 * first step that failed.  The general kernel reproduces it (pairSearchStep<..., EXACT>): an unflagged pair block also
 * gives the single step of c2 -- its letter counts, and the low two planes, which are the positions' own letters --
 * so when the pair's range comes out empty the range after c2 alone is computed from the same registers; if that one
 * is empty it is the reference's final range, otherwise the pattern died on c1 and the caller takes that one step
 * through the one-letter image.  The ordered (hits-only) search skips this and reports the pair's empty range; the
 * LF walk is exact either way.
 */
#ifndef AWFM_PAIR_H
#define AWFM_PAIR_H

#include "awfm_device.h"

namespace {

/* 24-bit count `c2` (0..3) out of the 96-bit string {w0, w1, w2} of a slice's second piece: bits 24 c2 .. 24 c2 + 23,
 * i.e. the funnel shift of the two dwords the count lies in (selects and one v_alignbit: written with 64-bit shifts
 * under `c2 < 2 ? .. : ..` hipcc made divergent branches of it, in every step of every kernel) */
__device__ __forceinline__ unsigned pairCount24(const Piece &h, unsigned c2) {
  const unsigned low = c2 < 2u ? h.y : (c2 == 2u ? h.z : h.w);
  const unsigned high = c2 < 2u ? h.z : (c2 == 2u ? h.w : 0u);
  const unsigned shift = (24u * c2) & 31u; /* 0, 24, 16, 8 */
  return __builtin_amdgcn_alignbit(high, low, shift) & kPairCountMask;
}

/* positions of a slice whose pair code is `pi`; pm[i] = bit i of pi as an all-ones mask */
__device__ __forceinline__ unsigned pairOccSlice(const Piece &planes, unsigned pm0, unsigned pm1, unsigned pm2, unsigned pm3) {
  return ~((planes.x ^ pm0) | (planes.y ^ pm1) | (planes.z ^ pm2) | (planes.w ^ pm3));
}

/* superblock base of entry `pi` (pair 0..15, or 16 + letter) at position q: from LDS (images below 2^32 positions) or
 * from memory.  LDS_STRIDE: entries per superblock of the LDS copy -- the search keeps the 16 pairs only (every KB of
 * LDS costs it resident workgroups), the walk all kPairSuperStride. */
template <bool NARROW, unsigned LDS_STRIDE = kPairSuperStride>
__device__ __forceinline__ typename PositionType<NARROW>::type pairSuperBase(const DevIndex &ix, const unsigned *sPairSuper,
                                                                            unsigned long long q, unsigned pi) {
  typedef typename PositionType<NARROW>::type pos_t;
  const unsigned sb = (unsigned)(q >> kPairSuperShift);
  if (NARROW) {
    /* two different loads, not one load through a selected pointer: hipcc otherwise merges them into a FLAT load,
     * which goes through the texture path even when the address is in LDS and makes the wait for it a wait for
     * every block read in flight.  The empty asm keeps the LDS read a ds_read of its own (it is waited for on the
     * spot, behind the block reads that are in flight anyway). */
    unsigned v;
    if (ix.pairSuperInLds) {
      v = sPairSuper[sb * LDS_STRIDE + pi];
      asm volatile("" : "+v"(v));
    } else {
      v = ix.pairSuper32[sb * kPairSuperStride + pi];
    }
    return (pos_t)v;
  }
  return (pos_t)ix.pairSuper[sb * kPairSuperStride + pi];
}

/* copies the pair tables into LDS: sPairC[16], and (NARROW) the first LDS_STRIDE 32-bit bases of every superblock */
template <bool NARROW, unsigned LDS_STRIDE = kPairSuperStride>
__device__ __forceinline__ void pairStageTables(const DevIndex &ix, unsigned long long *sPairC, unsigned *sPairSuper) {
  if (!ix.pairBlocks) return;
  if (threadIdx.x < 16) sPairC[threadIdx.x] = ix.pairC[threadIdx.x];
  if (NARROW && ix.pairSuperInLds)
    for (unsigned e = threadIdx.x; e < ix.numPairSuper * LDS_STRIDE; e += blockDim.x)
      sPairSuper[e] = ix.pairSuper32[(e / LDS_STRIDE) * kPairSuperStride + e % LDS_STRIDE];
}

/* what pairSearchStep did */
enum PairStep : unsigned {
  kPairStepped = 0u,      /* sp/ep are the range after both characters (EXACT: or the reference's final empty range) */
  kPairFlagged = 1u,      /* a block is flagged: sp/ep untouched, the caller takes both steps through the one-letter image */
  kPairDiedOnSecond = 2u, /* EXACT only: sp/ep are the non-empty range after c2; the step of c1, which empties it, is the caller's */
};

/*
 * Two backward steps (c2 first, then c1; pi = 4 c1 + c2) of a query by the 4 lanes of its group: lane k holds slice k
 * of a pair block.  Loads and rank are arranged as in nucFastStep.  EXACT: see the head of this file; sC = prefix sums.
 */
template <bool NARROW, bool EXACT = false>
__device__ __forceinline__ PairStep pairSearchStep(const DevIndex &ix, const unsigned long long *sPairC,
                                                   const unsigned *sPairSuper, const unsigned *sMask, unsigned slice,
                                                   unsigned pi, typename PositionType<NARROW>::type &sp,
                                                   typename PositionType<NARROW>::type &ep,
                                                   const unsigned long long *sC = nullptr) {
  typedef typename PositionType<NARROW>::type pos_t;
  const pos_t q0 = sp - 1, q1 = ep;
  const unsigned long long blk0 = q0 >> kBlockShift, blk1 = q1 >> kBlockShift;
  const bool same = blk0 == blk1;
  /* every lane fetches both pieces of its slice (fetching only what the rank reads -- plane pieces up to the
   * position, the count piece in the owning lane: 56 of 128 bytes -- was measured slower, 4.68 against 4.40 ms per
   * 10^8 random 21-mers: four predicated loads instead of two; so was reading only the count piece of c1, one dword
   * per lane, and assembling count and flag by a sum over the group: 5.1 against 4.45 ms random, 12.4 against 10.4
   * mixed lengths, 8.9 against 9.4 planted) */
  const Piece p0 = *(const Piece *)(ix.pairBlocks + pairPlanesAt(blk0, slice));
  const Piece h0 = *(const Piece *)(ix.pairBlocks + pairCountsAt(blk0, slice));
  Piece p1, h1;
  asm volatile("" : "=v"(p1), "=v"(h1));
  if (!same) {
    p1 = *(const Piece *)(ix.pairBlocks + pairPlanesAt(blk1, slice));
    h1 = *(const Piece *)(ix.pairBlocks + pairCountsAt(blk1, slice));
  }
  const unsigned pm0 = 0u - (pi & 1u), pm1 = 0u - ((pi >> 1) & 1u), pm2 = 0u - ((pi >> 2) & 1u), pm3 = 0u - (pi >> 3);
  const unsigned mask0 = sMask[((unsigned)q0 & kBlockMask) * kSlices + slice];
  const unsigned mask1 = sMask[((unsigned)q1 & kBlockMask) * kSlices + slice];
  const pos_t cPair = (pos_t)sPairC[pi];
  const pos_t super0 = pairSuperBase<NARROW, 16u>(ix, sPairSuper, q0, pi), super1 = pairSuperBase<NARROW, 16u>(ix, sPairSuper, q1, pi);
  const unsigned sameMask = same ? ~0u : 0u;
  const unsigned occ0 = pairOccSlice(p0, pm0, pm1, pm2, pm3), occ1 = pairOccSlice(p1, pm0, pm1, pm2, pm3);
  const unsigned n0 = __popc(occ0 & mask0);
  const unsigned n1 = __popc(__builtin_amdgcn_bitop3_b32(occ0, occ1, sameMask, 0xE4) & mask1); /* same ? occ0 : occ1 */
  asm volatile("" ::"v"(p0), "v"(p1), "v"(h0), "v"(h1));
  const unsigned c0 = pairCount24(h0, pi & 3u), c1 = pairCount24(h1, pi & 3u);
  const unsigned base0 = groupShfl<4>(c0, pi >> 2);
  unsigned base1 = groupShfl<4>(c1, pi >> 2);
  base1 = same ? base0 : base1;
  const unsigned flagged = (h0.x | (same ? 0u : h1.x)) >> 31; /* the flag is in every slice of a flagged block */
  const unsigned packed = groupSum<4>(n0 | (n1 << 16));
  if (flagged) return kPairFlagged;
  const pos_t sp2 = cPair + super0 + (pos_t)base0 + (pos_t)(packed & 0xFFFFu);
  const pos_t ep2 = cPair + super1 + (pos_t)base1 + (pos_t)(packed >> 16) - (pos_t)1;
  if (EXACT && sp2 > ep2) { /* group-uniform; once per query at most */
    /* the step of c2 alone (ref src/AwFmSearch.c:42-103) out of the same pieces: letter count of slice c2's count
     * piece + positions whose own letter (planes 0 and 1) is c2 */
    const unsigned c2 = pi & 3u;
    const unsigned lm0 = 0u - (c2 & 1u), lm1 = 0u - (c2 >> 1);
    const unsigned own0 = ~((p0.x ^ lm0) | (p0.y ^ lm1));
    const unsigned own1 = same ? own0 : ~((p1.x ^ lm0) | (p1.y ^ lm1));
    const unsigned ranks = groupSum<4>(__popc(own0 & mask0) | (__popc(own1 & mask1) << 16));
    const unsigned letters0 = groupShfl<4>(h0.x, c2);
    unsigned letters1 = groupShfl<4>(h1.x, c2);
    letters1 = same ? letters0 : letters1;
    const unsigned long long e0 = (unsigned long long)(q0 >> kPairSuperShift) * kPairSuperStride + 16u + c2;
    const unsigned long long e1 = (unsigned long long)(q1 >> kPairSuperShift) * kPairSuperStride + 16u + c2;
    const pos_t ls0 = NARROW ? (pos_t)ix.pairSuper32[e0] : (pos_t)ix.pairSuper[e0];
    const pos_t ls1 = NARROW ? (pos_t)ix.pairSuper32[e1] : (pos_t)ix.pairSuper[e1];
    const pos_t cLetter = (pos_t)sC[c2];
    sp = cLetter + ls0 + (pos_t)letters0 + (pos_t)(ranks & 0xFFFFu);
    ep = cLetter + ls1 + (pos_t)letters1 + (pos_t)(ranks >> 16) - (pos_t)1;
    return sp > ep ? kPairStepped : kPairDiedOnSecond;
  }
  sp = sp2;
  ep = ep2;
  return kPairStepped;
}

}  // namespace

#endif
