/*
 * awfm_locate_kernel.h -- suffix-array backtrace: BWT position of a hit -> text position, two kernels.
 *
 * walkKernel: one hit at a time per group of G lanes (same slice ownership as the search kernel: lane j holds
 *   slices j*S..j*S+S-1 of a block, S = 4/G); a group takes its hits in batches of 4*G consecutive entries,
 *   read and written as whole lines.  Per LF step the group reads the block of the current BWT
 *   position (one 64-B granule, amino one 128-B line), extracts the letter stored there (the owning lane builds
 *   the code from its plane words, the group gets it by ds_bpermute), ranks that letter up to the position and continues
 *   at C[a] + Occ(a,p) - 1 until the position is sampled.  Persistent with refill: a group that reaches a
 *   sampled position stores {sample index, steps walked} in place and continues with its next hit in the
 *   same iteration, so every iteration is exactly one dependent block read per group and no lane waits
 *   for the longest chain of its wave (chain lengths are roughly geometric with mean ratio-1).
 * finishKernel: one thread per hit reads the bit-packed sample and stores
 *   (SA_s[sample] + steps) mod bwtLength.
 * Keeping the sample read out of the walk loop matters because the loop is VALU-issue bound: with both
 * paths in one loop every iteration executed both (some group of the wave is always at a sample).
 *
 * ref src/AwFmParallelSearch.c:338-361, src/AwFmSearch.c:369-427, src/AwFmOccurrence.c:170-217,
 *     src/AwFmIndexStruct.c:88-91, src/AwFmSuffixArray.c:22-39, :114-142, :179-191.
 */
#ifndef AWFM_LOCATE_KERNEL_H
#define AWFM_LOCATE_KERNEL_H

#include "awfm_search_kernel.h"
#include "awfm_pair.h"

namespace {

/* in-place hand-over between the two kernels: bit 63 set, steps in bits 62..40, sample index in 39..0 */
/* a walk given up after stepCap steps (the construction of the full suffix array): bit 62 set, the steps walked in bits
 * 61..40 (a cap is 32 x ratio: 13 bits), the BWT position it stands at in bits 39..0 -- untagged for finishKernel, which
 * passes it on; awfm_gpu.hip completes such entries from each other */
constexpr unsigned long long kWalkParked = 1ull << 62;
constexpr unsigned long long kWalkTag = 1ull << 63;
constexpr unsigned kWalkStepBits = 23;
constexpr unsigned long long kWalkSampleMask = (1ull << 40) - 1ull;

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

/* (SA_s[sample] + steps) mod bwtLength (ref src/AwFmSuffixArray.c:183-191); the sum is below 2*bwtLength
 * for a consistent index, the second reduction only guards a corrupt one */
__device__ __forceinline__ unsigned long long finishPosition(const DevIndex &ix, unsigned long long sample,
                                                             unsigned long long steps) {
  unsigned long long v = sampledSaValue(ix, sample) + steps;
  if (v >= ix.bwtLength) v -= ix.bwtLength;
  if (v >= ix.bwtLength) v %= ix.bwtLength;
  return v;
}

/* One LF step by ONE thread (ref src/AwFmSearch.c:369-427, src/AwFmOccurrence.c:170-217): the slow path that takes a walk on
 * from where walkKernel gave it up (finishKernel).  Nothing from LDS: the superblock bases and the prefix sums are read where
 * they live; `amino`: the image's alphabet. */
__device__ inline unsigned long long lfStepSerial(const DevIndex &ix, const bool amino, const unsigned long long p) {
  const unsigned long long blk = p >> kBlockShift;
  const unsigned local = (unsigned)p & kBlockMask, slice = local >> 5, bit = local & 31u;
  if (!amino) {
    Piece pc[kSlices];
    for (unsigned k = 0; k < kSlices; k++) pc[k] = *(const Piece *)(ix.blocks + (blk * kSlices + k));
    unsigned code = 0;
    for (unsigned k = 0; k < kSlices; k++)
      if (k == slice) code = ((pc[k].x >> bit) & 1u) | (((pc[k].y >> bit) & 1u) << 1) | (((pc[k].z >> bit) & 1u) << 2);
    const unsigned letter = (0x00152435u >> (4u * code)) & 7u; /* code -> index {5,3,4,2,5,1,0,0}, ref src/AwFmLetter.c:49-53 */
    if (letter == 5u) return 0ull;                             /* sentinel: ref src/AwFmSearch.c:384-386 */
    const unsigned safe = letter < 5u ? letter : 0u;
    const PlaneSel3 sel = nucPlaneSel(safe);
    unsigned n = 0;
    unsigned long long acgt = 0, mine = 0;
    for (unsigned k = 0; k < kSlices; k++) {
      n += __popc(nucOccSlice(pc[k], sel) & sliceMask(local, k));
      acgt += pc[k].w;
      if (k == safe) mine = pc[k].w;
    }
    const unsigned long long sb = blk >> (ix.nucSuperShift - kBlockShift);
    unsigned long long base;
    if (safe < 4u) {
      base = mine + ix.super[sb * 4ull + safe];
    } else { /* X: everything before the block that is not a,c,g,t or the sentinel */
      acgt += ix.super[sb * 4ull] + ix.super[sb * 4ull + 1ull] + ix.super[sb * 4ull + 2ull] + ix.super[sb * 4ull + 3ull];
      const unsigned long long before = blk << kBlockShift;
      base = before - acgt - (ix.sentinelPos < before ? 1ull : 0ull);
    }
    return ix.prefixSums[safe] + base + n - 1ull;
  }
  Piece lo[kSlices], hi[kSlices];
  for (unsigned k = 0; k < kSlices; k++) {
    lo[k] = *(const Piece *)(ix.blocks + (blk * kSlices + k) * 2ull);
    hi[k] = *(const Piece *)(ix.blocks + (blk * kSlices + k) * 2ull + 1ull);
  }
  unsigned code = 0;
  for (unsigned k = 0; k < kSlices; k++)
    if (k == slice)
      code = ((lo[k].x >> bit) & 1u) | (((lo[k].y >> bit) & 1u) << 1) | (((lo[k].z >> bit) & 1u) << 2) | (((lo[k].w >> bit) & 1u) << 3) |
             (((hi[k].x >> bit) & 1u) << 4);
  const unsigned letter = kAminoTables.letterOfCode[code];
  if (letter == 21u) return 0ull; /* sentinel: ref src/AwFmSearch.c:414-416 */
  const unsigned safe = letter < 21u ? letter : 0u;
  const unsigned pm = kAminoTables.planeMask[safe];
  unsigned n = 0, count16 = 0;
  for (unsigned k = 0; k < kSlices; k++) {
    n += __popc(aminoOccSlice(lo[k], hi[k], pm & 0xFFu, pm >> 8) & sliceMask(local, k));
    if (k == safe / 6u) count16 = aminoCount16(hi[k], safe % 6u);
  }
  return ix.prefixSums[safe] + ix.super[(p >> kAminoSuperShift) * kAminoSuperStride + safe] + count16 + n - 1ull;
}

/* workgroup size: 512 threads for the pair variant, whose LDS copy of the pair image's superblock bases (24 KB for a
 * GRCh38-sized index) would otherwise limit a CU to 6 workgroups of 256 */
constexpr int walkThreads(bool pair) { return pair ? 512 : kThreads; }

/* PERLANE: hits a lane holds of its group's batch (a batch = PERLANE * G hits, walked one after the other).  4: a batch is
 * read and written as whole lines, which is what a long hit list wants; 1: a quarter of the chain per group, which is what
 * a short one wants (7 * 10^4 hits of 10^8 random 21-mers: every group of the grid gets at most one batch either way, and
 * the kernel's time is the length of that chain: 0.16 ms with batches of 16) */
template <bool AMINO, int G, bool POW2, bool NARROW, bool PAIR = false, unsigned PERLANE = 4u>
__global__ void __launch_bounds__(walkThreads(PAIR)) __attribute__((amdgpu_num_sgpr(80)))
    walkKernel(const DevIndex ix, unsigned long long totalHits, unsigned long long *__restrict__ positions,
               const unsigned long long *__restrict__ totalOnDevice = nullptr, const unsigned stepCap = 0u,
               const unsigned giveUpAfter = (1u << kWalkStepBits) - 1u) {
  static_assert(!PAIR || (!AMINO && G == 4), "pair steps: nucleotide images, 4 lanes per hit");
  /* the number of hits may still be on the device when the kernel is launched (awfmGpuLocateOnDevice: the total of the
   * scan, never read by the host); totalHits is then the capacity of `positions` */
  if (totalOnDevice) {
    const unsigned long long t = *totalOnDevice;
    totalHits = t < totalHits ? t : totalHits;
  }
  constexpr int S = (int)kSlices / G;
  constexpr int V = AMINO ? 2 : 1;
  constexpr int kGroups = walkThreads(PAIR) / G;
  typedef typename PositionType<NARROW>::type pos_t;
  __shared__ unsigned long long sC[24];
  __shared__ AminoShared sAmino;
  __shared__ unsigned long long sSuper[!AMINO && !NARROW ? kMaxNucSuper * 4 : 1];
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  /* PAIR: two LF steps per block read through the pair image (awfm_pair.h) wherever the position's block is not
   * flagged; a flagged block's position goes on, in the same iteration, to a single step through the one-letter
   * image */
  __shared__ unsigned long long sPairC[PAIR ? 16 : 1];
  extern __shared__ unsigned sPairSuper[];
  if (AMINO) aminoStageTables(sAmino);
  if (!AMINO) nucStageSuper<NARROW>(ix, sSuper);
  if (PAIR) pairStageTables<NARROW>(ix, sPairC, sPairSuper);
  __syncthreads();
  const unsigned gl = threadIdx.x % G;
  const unsigned firstSlice = gl * S;
  const unsigned long long numGroups = (unsigned long long)gridDim.x * kGroups;
  const pos_t ratio = (pos_t)ix.saRatio;
  /* stepCap (the construction of the full suffix array, awfm_gpu.hip): a position that has not reached a sample after so
   * many steps is given up -- parked: kWalkParked, where it stands and how far it came -- rather than followed for up to 2^23 steps: in a text
   * with R long runs of one letter the suffixes inside the runs map, LF step by LF step, R places further in the suffix
   * array, and with R a multiple of the sampling ratio they never meet a sample until a run ends */
  /* without a cap (every ordinary locate): the hand-over holds 23 bits of steps, so a walk that has not met a sample after
   * `giveUpAfter` (2^23 - 1) steps -- a hit right behind such a run: rare, but a valid index -- is parked too, in its own
   * format (kWalkParked | one step more than giveUpAfter << 61 | position), and finishKernel walks it to its sample one
   * thread at a time: slow, and exact (round 4 finished such a walk as if it stood on a sample) */
  const unsigned long long maxSteps = stepCap ? (unsigned long long)stepCap : (unsigned long long)giveUpAfter;

  /* A group works through batches of 4*G consecutive hits (lane j holds hits 4j..4j+3 of the batch, 32 B):
   * one coalesced read brings a batch in, the hand-over values replace the BWT positions in the registers,
   * one coalesced write takes the batch out.  (One hit per refill, 8 bytes at a time, cost an extra
   * partial-line read and a read-modify-write per hit: the L2 lines do not survive between the refills.) */
  constexpr unsigned kPerLane = PERLANE, kBatch = kPerLane * G;
  static_assert(PERLANE == 4u || PERLANE == 1u, "a lane holds four hits of a batch, or one");
  const unsigned long long batchStride = numGroups * kBatch;
  const bool aligned = ((unsigned long long)positions & 15ull) == 0ull;
  /* (the four entries of a lane are named scalars: hipcc moves a small array that is indexed in any
   * non-constant way to LDS) */
  struct Four {
    unsigned long long a, b, c, d;
  };
  auto loadBatch = [&](unsigned long long base) -> Four {
    const unsigned long long first = base + kPerLane * gl;
    Four r;
    if (kPerLane == 1u) {
      r.a = first < totalHits ? positions[first] : 0ull;
      r.b = r.c = r.d = 0ull;
    } else if (aligned && first + kPerLane <= totalHits) {
      const ulonglong2 lo = *(const ulonglong2 *)(positions + first), hi = *(const ulonglong2 *)(positions + first + 2);
      r.a = lo.x;
      r.b = lo.y;
      r.c = hi.x;
      r.d = hi.y;
    } else {
      r.a = first < totalHits ? positions[first] : 0ull;
      r.b = first + 1 < totalHits ? positions[first + 1] : 0ull;
      r.c = first + 2 < totalHits ? positions[first + 2] : 0ull;
      r.d = first + 3 < totalHits ? positions[first + 3] : 0ull;
    }
    return r;
  };
  auto storeBatch = [&](unsigned long long base, const Four &v) {
    const unsigned long long first = base + kPerLane * gl;
    if (kPerLane == 1u) {
      if (first < totalHits) positions[first] = v.a;
    } else if (aligned && first + kPerLane <= totalHits) {
      *(ulonglong2 *)(positions + first) = make_ulonglong2(v.a, v.b);
      *(ulonglong2 *)(positions + first + 2) = make_ulonglong2(v.c, v.d);
    } else {
      if (first < totalHits) positions[first] = v.a;
      if (first + 1 < totalHits) positions[first + 1] = v.b;
      if (first + 2 < totalHits) positions[first + 2] = v.c;
      if (first + 3 < totalHits) positions[first + 3] = v.d;
    }
  };
  /* hit j of the batch, for every lane of the group (32-bit positions travel as one word) */
  auto pick = [&](const Four &v, unsigned j) -> pos_t {
    const unsigned k = j % kPerLane;
    const unsigned long long mine = k == 0u ? v.a : (k == 1u ? v.b : (k == 2u ? v.c : v.d));
    const unsigned lo = groupShfl<G>((unsigned)mine, j / kPerLane);
    if (NARROW) return (pos_t)lo;
    return (pos_t)(((unsigned long long)groupShfl<G>((unsigned)(mine >> 32), j / kPerLane) << 32) | lo);
  };

  unsigned long long batchBase = (((unsigned long long)blockIdx.x * walkThreads(PAIR) + threadIdx.x) / G) * kBatch;
  bool alive = batchBase < totalHits;
  Four slot = {0ull, 0ull, 0ull, 0ull}, nslot = {0ull, 0ull, 0ull, 0ull};
  unsigned cnt = 0, j = 0;
  if (alive) {
    slot = loadBatch(batchBase);
    cnt = totalHits - batchBase < kBatch ? (unsigned)(totalHits - batchBase) : kBatch;
  }
  if (batchBase + batchStride < totalHits) nslot = loadBatch(batchBase + batchStride);
  pos_t p = pick(slot, 0u);
  unsigned steps = 0;
  while (alive) {
    bool sampled = POW2 ? (p & (ratio - 1)) == 0 : (p % ratio) == 0; /* ref src/AwFmIndexStruct.c:88-91 */
    if (sampled || steps >= maxSteps) {
      /* hand the hit over, or park it */
      const unsigned long long sample = POW2 ? (unsigned long long)(p >> ix.saShift) : (unsigned long long)(p / ratio);
      const unsigned long long result = sampled ? (kWalkTag | ((unsigned long long)steps << 40) | (sample & kWalkSampleMask))
                                                : (stepCap ? (kWalkParked | ((unsigned long long)steps << 40) | ((unsigned long long)p & kWalkSampleMask))
                                                           : (kWalkParked | ((unsigned long long)(steps - (unsigned)maxSteps) << 61) | (unsigned long long)p));
      const bool owner = gl == j / kPerLane;
      const unsigned k = j % kPerLane;
      slot.a = owner && k == 0u ? result : slot.a;
      slot.b = owner && k == 1u ? result : slot.b;
      slot.c = owner && k == 2u ? result : slot.c;
      slot.d = owner && k == 3u ? result : slot.d;
      j++;
      if (j == cnt) {
        storeBatch(batchBase, slot);
        batchBase += batchStride;
        alive = batchBase < totalHits;
        slot = nslot;
        cnt = !alive ? 0u : (totalHits - batchBase < kBatch ? (unsigned)(totalHits - batchBase) : kBatch);
        j = 0;
        if (batchBase + batchStride < totalHits) nslot = loadBatch(batchBase + batchStride);
      }
      p = pick(slot, j);
      steps = 0;
      sampled = POW2 ? (p & (ratio - 1)) == 0 : (p % ratio) == 0;
    }
    bool walk = alive && !sampled; /* a refilled hit that is sampled right away is handed over next iteration */
    if (PAIR) {
      /* The pair block of p gives both LF(p) (its low code bits are the letters, one count per letter) and LF(LF(p)):
       * the hit moves two steps unless the position in between is a sampled one, where it moves one and ends there in
       * the next iteration.  Flagged blocks (ambiguity letter or sentinel around) go on to the one-letter step below. */
      const unsigned long long pblk = (unsigned long long)(p >> kBlockShift);
      const unsigned plocal = (unsigned)p & kBlockMask;
      Piece pl = (Piece)(0u), ph = (Piece)(0u);
      if (walk) {
        pl = *(const Piece *)(ix.pairBlocks + pairPlanesAt(pblk, gl));
        ph = *(const Piece *)(ix.pairBlocks + pairCountsAt(pblk, gl));
      }
      __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0) */
      if (walk && (ph.x >> 31) == 0u) {
        const unsigned bit = plocal & 31u, ownerSlice = plocal >> 5;
        const unsigned mine = ((pl.x >> bit) & 1u) | (((pl.y >> bit) & 1u) << 1) | (((pl.z >> bit) & 1u) << 2) | (((pl.w >> bit) & 1u) << 3);
        const unsigned pi = groupShfl<G>(mine, ownerSlice) & 15u; /* pair code at p */
        const unsigned c2 = pi & 3u;                            /* the letter at p */
        const unsigned mask = sliceMask(plocal, gl);
        const unsigned m0 = 0u - (pi & 1u), m1 = 0u - ((pi >> 1) & 1u), m2 = 0u - ((pi >> 2) & 1u), m3 = 0u - (pi >> 3);
        const unsigned sameLetter = ~((pl.x ^ m0) | (pl.y ^ m1));
        const unsigned nLetter = __popc(sameLetter & mask), nPair = __popc(sameLetter & ~((pl.z ^ m2) | (pl.w ^ m3)) & mask);
        const unsigned totals = groupSum<G>(nLetter | (nPair << 16));
        const unsigned letterBase = groupShfl<G>(ph.x & kPairCountMask, c2);
        const unsigned pairBase = groupShfl<G>(pairCount24(ph, pi & 3u), pi >> 2);
        const pos_t one = (pos_t)sC[c2] + pairSuperBase<NARROW>(ix, sPairSuper, (unsigned long long)p, 16u + c2) + (pos_t)letterBase +
                          (pos_t)(totals & 0xFFFFu) - (pos_t)1; /* LF(p), ref src/AwFmSearch.c:369-392 */
        const pos_t two = (pos_t)sPairC[pi] + pairSuperBase<NARROW>(ix, sPairSuper, (unsigned long long)p, pi) + (pos_t)pairBase +
                          (pos_t)(totals >> 16) - (pos_t)1; /* LF(LF(p)) */
        const bool stop = POW2 ? (one & (ratio - 1)) == 0 : (one % ratio) == 0;
        p = stop ? one : two;
        steps += stop ? 1u : 2u;
        walk = false;
      }
    }
    const unsigned long long blk = (unsigned long long)(p >> kBlockShift);
    const unsigned local = (unsigned)p & kBlockMask;
    Piece pc[S][V];
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
      for (int v = 0; v < V; v++) pc[s][v] = (Piece)(0u);
    if (walk) {
      const Piece *at = (const Piece *)(ix.blocks + (blk * kSlices + firstSlice) * V);
#pragma unroll
      for (int s = 0; s < S; s++)
#pragma unroll
        for (int v = 0; v < V; v++) pc[s][v] = at[s * V + v];
    }
    __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0): the single drain of the iteration */
    if (walk) {
      const unsigned bit = local & 31u, ownerSlice = local >> 5;
      /* letter stored at p: code bits from the owning slice */
      unsigned myCode = 0;
#pragma unroll
      for (int s = 0; s < S; s++) {
        unsigned code = ((pc[s][0].x >> bit) & 1u) | (((pc[s][0].y >> bit) & 1u) << 1) | (((pc[s][0].z >> bit) & 1u) << 2);
        if (AMINO) code |= (((pc[s][0].w >> bit) & 1u) << 3) | (((pc[s][V - 1].x >> bit) & 1u) << 4);
        myCode = (ownerSlice % S) == (unsigned)s ? code : myCode;
      }
      const unsigned code = groupShfl<G>(myCode, ownerSlice / S);
      pos_t next;
      if (AMINO) {
        const unsigned letter = sAmino.letterOfCode[code];
        const unsigned safe = letter < 21u ? letter : 0u;
        /* the superblock base of the letter just read: a second, dependent read, but into a table that stays in
         * the L2 (24 words per 2^16 positions) */
        const unsigned long long super = ix.super[(unsigned long long)(p >> kAminoSuperShift) * kAminoSuperStride + safe];
        const unsigned pm = sAmino.planeMask[safe];
        const unsigned ones = pm & 0xFFu, zeros = pm >> 8;
        unsigned n = 0, mine = 0;
        const unsigned slice = safe / 6u, sub = safe % 6u;
#pragma unroll
        for (int s = 0; s < S; s++) {
          n += __popc(aminoOccSlice(pc[s][0], pc[s][V - 1], ones, zeros) & sliceMask(local, firstSlice + s));
          mine = (slice % S) == (unsigned)s ? aminoCount16(pc[s][V - 1], sub) : mine;
        }
        next = (pos_t)sC[safe] + (pos_t)super + (pos_t)groupShfl<G>(mine, slice / S) + (pos_t)groupSum<G>(n) - (pos_t)1;
        if (letter == 21u) next = 0; /* sentinel: ref src/AwFmSearch.c:414-416 */
      } else {
        const unsigned letter = (0x00152435u >> (4u * code)) & 7u; /* code -> index {5,3,4,2,5,1,0,0}, ref src/AwFmLetter.c:49-53 */
        const unsigned safe = letter < 5u ? letter : 0u;
        const PlaneSel3 sel = nucPlaneSel(safe);
        unsigned n = 0;
        Piece mine[S];
#pragma unroll
        for (int s = 0; s < S; s++) {
          mine[s] = pc[s][0];
          n += __popc(nucOccSlice(pc[s][0], sel) & sliceMask(local, firstSlice + s));
        }
        next = (pos_t)sC[safe] + nucBaseAny<G, NARROW>(ix, sSuper, mine, safe, blk) + (pos_t)groupSum<G>(n) - (pos_t)1;
        if (letter == 5u) next = 0; /* sentinel: ref src/AwFmSearch.c:384-386 */
      }
      p = next;
      steps++;
    }
  }
}

/* second half: the bit-packed sample of every handed-over hit */
/* out == positions: in place.  Otherwise every entry is written to `out`, which may be page-locked host memory (the
 * pipeline of awfm_gpu_stream.hip lets the kernel that produces the positions deliver them: sequential 8-byte stores) */
/* resume: 0 -- parked entries are passed on as they are (the construction of the full suffix array completes them from each
 * other); 1 / 2 -- a nucleotide / amino image's ordinary locate: an entry walkKernel parked after giveUpAfter (+ 0 or 1)
 * steps is walked on here, one LF step at a time by its thread, until it stands on a sample (ref
 * src/AwFmParallelSearch.c:338-361 walks every hit that way) */
__global__ void finishKernel(const DevIndex ix, unsigned long long totalHits, const unsigned long long *positions,
                             unsigned long long *out, const unsigned long long *__restrict__ totalOnDevice = nullptr,
                             const unsigned resume = 0u, const unsigned giveUpAfter = (1u << kWalkStepBits) - 1u) {
  if (totalOnDevice) {
    const unsigned long long t = *totalOnDevice;
    totalHits = t < totalHits ? t : totalHits;
  }
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  const bool inPlace = out == positions;
  for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < totalHits; t += stride) {
    const unsigned long long v = positions[t];
    if (v & kWalkTag) {
      out[t] = finishPosition(ix, v & kWalkSampleMask, (v >> 40) & ((1ull << kWalkStepBits) - 1ull));
    } else if ((v & kWalkParked) && resume) {
      unsigned long long p = v & ((1ull << 61) - 1ull), steps = (unsigned long long)giveUpAfter + ((v >> 61) & 1ull);
      while (p % ix.saRatio != 0ull && steps <= ix.bwtLength) { /* (the bound: a corrupt index must not hang the device) */
        p = lfStepSerial(ix, resume == 2u, p);
        steps++;
      }
      out[t] = finishPosition(ix, p / ix.saRatio, steps);
    } else if (!inPlace) {
      out[t] = v;
    }
  }
}

}  // namespace

#endif
