/*
 * awfm_ordered_kernel.h -- the ordered, hits-only search of large nucleotide batches, fixed-length or CSR
 * (awfmGpuSearchHits).
 *
 * The backward search of a batch of unrelated k-mers reads BWT blocks at random: every step of every query is
 * one 128-B line from somewhere in the image, and the chip delivers about 5*10^10 such lines per second
 * whatever the kernel does.  The order of the queries inside a batch is not part of the
 * result, so this path picks the order that makes those reads local:
 *
 *   1. a counting pass (encodeCodes4Kernel / encodeRecordsKernel): the k-mers as 2-bit codes and a histogram of the 2048
 *      buckets the leading bits of the string their search starts from fall into (its table index; the k-mer itself when it is
 *      shorter than the seed).  K-mers with ambiguity characters, no characters or more than 32 go to a last bin and are
 *      searched by the general kernel afterwards (searchKernel<INDIRECT>).
 *   2. a partition pass (partitionKernel / partitionRecordsKernel): records in bucket order, any order inside a bucket.
 *   3. orderedSearchKernel: the same table lookup and the same backward steps as searchKernel (nucFastStep) over
 *      the records in that order.  Neighbours in it start in neighbouring table entries and, step after
 *      step, land in neighbouring blocks (the range of cP lies in the c-section of the BWT in the order of P),
 *      so most block reads hit the L2.  Each XCD has its own L2: workgroup b runs on XCD b % 8, every XCD walks
 *      one contiguous eighth of the order, and its waves take consecutive chunks from ticket counters so that
 *      what is in flight stays within what the L2 holds.
 *   Batches most of whose k-mers end at the deeper table skip the ordering: lookupSearchKernel (below) looks every k-mer's
 *   entry up in input order and searches the few that are still alive itself.
 *
 * Putting results back under the original query numbers is a scatter of one partial line per query, which costs
 * as much as the ordering saves -- unless it is sparse.  This path therefore reports HITS: the
 * output arrays are first filled with "no hit" ({1,0} / 0) by a streaming kernel, and only queries whose final
 * range is non-empty store it.  For a query without hits the reference's batch API reports count 0 and nothing
 * else, so nothing is lost; the exact empty range the stepping ended in is what awfmGpuSearch returns.
 *
 * Results of queries with hits are those of the reference algorithm (same table entry, same steps):
 * ref src/AwFmParallelSearch.c:222-313, src/AwFmKmerTable.c:4-51, src/AwFmSearch.c:42-159.
 */
#ifndef AWFM_ORDERED_KERNEL_H
#define AWFM_ORDERED_KERNEL_H

#include "awfm_search_kernel.h"
#include "awfm_pair.h"

namespace {

/* 4 characters -> 4 two-bit codes (first character in bits 7..6) and 4 "not a,c,g,t,u" flags (bit i = character i);
 * same SWAR decode as the window decode of searchKernel */
__device__ __forceinline__ void decodeWord(unsigned word, unsigned &packed, unsigned &badBits) {
  unsigned t = (word >> 1) & 0x03030303u;
  t ^= (t >> 1) & 0x01010101u;
  const unsigned b0 = t & 0x01010101u, b1 = (t >> 1) & 0x01010101u, b01 = b0 & b1;
  const unsigned expect = 0x61616161u + (b0 << 1) + b1 * 6u + b01 * 11u; /* 'a','c','g','t' */
  unsigned diff = ((word | 0x20202020u) ^ expect) & ~b01;
  diff |= diff >> 4;
  diff |= diff >> 2;
  diff |= diff >> 1;
  diff &= 0x01010101u;
  badBits = (diff & 1u) | ((diff >> 7) & 2u) | ((diff >> 14) & 4u) | ((diff >> 21) & 8u);
  packed = ((t & 3u) << 6) | ((t >> 4) & 0x30u) | ((t >> 14) & 0x0Cu) | (t >> 24);
}

/* The same for a kernel that only asks whether ANY character of a k-mer is not a,c,g,t,u: the expected letters come from
 * one byte permute ('a','c','g','t' selected by the codes), and the differences of the characters that count
 * (`countMask`: 0xFF per character of the k-mer in this word) are OR-ed into `diffs` as they are.  decodeWord: 35 VALU
 * instructions per word, this: 19 -- encodeCodes4Kernel was bound by its VALU instructions (94 % of its cycles), not by its
 * 2.9 GB. */
__device__ __forceinline__ void decodeWordAny(unsigned word, unsigned countMask, unsigned &packed, unsigned &diffs) {
  unsigned t = (word >> 1) & 0x03030303u;
  t ^= (t >> 1) & 0x01010101u;
  const unsigned b01 = t & (t >> 1) & 0x01010101u;                   /* code 3: 't' or 'u' (they differ in bit 0) */
  const unsigned expect = __builtin_amdgcn_perm(0u, 0x74676361u, t); /* byte i = "acgt"[code i] */
  diffs |= ((word | 0x20202020u) ^ expect) & ~b01 & countMask;
  packed = ((t & 3u) << 6) | ((t >> 4) & 0x30u) | ((t >> 14) & 0x0Cu) | (t >> 24);
}

/* 2-bit codes (last character in bits 1..0) and ambiguity mask (bit i = character i is not a,c,g,t,u) of the `len`
 * (1..32) ASCII characters at chars + start: the aligned dwords that hold them, a dword only read when it contains one */
__device__ __forceinline__ void decodeKmer(const unsigned char *__restrict__ chars, unsigned long long start, unsigned len,
                                           unsigned long long &codes, unsigned &bad) {
  const unsigned long long at = (unsigned long long)chars + start;
  /* (an integer turned pointer is a generic one, read by FLAT loads: say that it is global memory) */
  typedef const unsigned __attribute__((address_space(1))) *GlobalWords;
  const GlobalWords first = (GlobalWords)(at & ~3ull);
  const unsigned shift = (unsigned)at & 3u;
  const unsigned numDwords = (shift + len + 3u) >> 2;
  unsigned dw[9];
#pragma unroll
  for (int j = 0; j < 9; j++) dw[j] = (unsigned)j < numDwords ? first[j] : 0u;
  codes = 0;
  bad = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    unsigned packed = 0, badBits = 0;
    if (4u * j < len) decodeWord(__builtin_amdgcn_alignbyte(dw[j + 1], dw[j], shift), packed, badBits);
    codes = (codes << 8) | packed;
    bad |= badBits << (4 * j);
  }
  codes >>= 2u * (32u - len); /* character 0 was in bits 63..62: now the last character is in bits 1..0 */
  bad &= len >= 32u ? ~0u : ((1u << len) - 1u);
}

/* leading 15 bits of the string the search of a k-mer starts from: its last `depth` characters (the table index), or the
 * whole k-mer when it starts from a letter range -- where its search lands in the BWT */
__device__ __forceinline__ unsigned startKey15(unsigned long long codes, unsigned len, unsigned depth) {
  const unsigned span = depth ? depth : len; /* characters of that string, 1..32 */
  const unsigned long long str = span >= 32u ? codes : (codes & ((1ull << (2u * span)) - 1ull));
  return 2u * span >= 15u ? (unsigned)(str >> (2u * span - 15u)) : (unsigned)(str << (15u - 2u * span));
}

/* "no hit" everywhere: the ordered search only stores the queries that have hits */
__global__ void __launch_bounds__(256)
    fillNoHitKernel(ulonglong2 *__restrict__ ranges, unsigned *__restrict__ counts, const unsigned long long n) {
  const unsigned long long stride = (unsigned long long)gridDim.x * 256ull;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i < n; i += stride) {
    if (ranges) ranges[i] = make_ulonglong2(1ull, 0ull);
    if (counts) counts[i] = 0u;
  }
}

/* where the search of a k-mer of `len` characters starts: the deeper device-only table, the index's seed table,
 * or (0) the letter range of its last character (ref src/AwFmSearch.c:485-520: k-mers shorter than the seed) */
__host__ __device__ inline unsigned orderStartDepth(unsigned len, unsigned seedK, unsigned deepK) {
  if (deepK != 0u && len >= deepK) return deepK;
  return len >= seedK ? seedK : 0u;
}

/*
 * ---- the hand-written ordering of fixed-length batches: one counting pass + one partition pass ----
 *
 * The order the search needs is coarse: what is in flight on an XCD must stay within what its L2 holds, and with the k-mers
 * partitioned by the leading 11 bits of their table index (2048 buckets, any order inside a bucket) orderedSearchKernel
 * takes 4.59 instead of 4.41 ms per 10^8 random 21-mers (12 bits 4.54, 10 bits 4.83, measured on round 3's
 * sorted path).  A 2048-way partition is ONE pass whose stores still leave a workgroup as runs of 64 bytes, where the
 * 15-bit order took two radix-sort passes over (key, record) pairs after an encoding pass:
 *   encodeCodesKernel   k-mer characters -> 2-bit codes (8 B per k-mer, original order) + bucket histogram (skipped for
 *                       bit-packed input, which is its own code array: then the kernel only counts);
 *   bucketScanKernel    exclusive scan of the 2049 bin counts (the last bin: k-mers left to the general kernel);
 *   partitionKernel     every workgroup takes tiles of 16384 codes, ranks them inside the tile by LDS atomics, reserves
 *                       one run per non-empty bucket with a global atomic, and writes the 8-byte records run by run.
 * A record is {rest of the code string : 64 - indexBits, query number : indexBits}; the bucket's own bits are not stored:
 * the search kernel knows the bucket of a record from its position (bucketStart, tracked per wave).
 * 10^8 21-mers: 2.9 + 1.6 GB of memory traffic instead of 3.1 + 0.2 + 2 x 2.0 GB.
 */
constexpr unsigned kBucketBitsMax = 11; /* 2048 buckets (+ 1 for the k-mers left to the general kernel) */
constexpr unsigned kPartitionThreads = 1024, kPartitionItems = 16, kPartitionTile = kPartitionThreads * kPartitionItems;
constexpr unsigned long long kCodeGeneral = 1ull << 63; /* a code word of a k-mer the ordered kernel does not cover */
/* counters[bin] += 1 for the lanes with `valid`, returning each one's rank.  A batch that arrives sorted puts every lane of a
 * wave -- every k-mer of a tile -- into the same bin: 64 LDS atomics on one word per instruction, serialised (a sorted
 * batch took 2.9 instead of 1.0 ms through the two ordering passes).  When the valid lanes agree on the bin one of them adds
 * for all. */
__device__ __forceinline__ unsigned ldsCountRank(unsigned *counters, unsigned bin, bool valid) {
  const unsigned long long vm = __ballot(valid);
  if (vm == 0ull) return 0u; /* wave-uniform */
  const int leader = __ffsll((long long)vm) - 1;
  const unsigned bin0 = (unsigned)__shfl((int)bin, leader, 64);
  if (__ballot(valid && bin == bin0) == vm) { /* wave-uniform */
    unsigned base = 0;
    if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(&counters[bin0], (unsigned)__popcll(vm));
    base = (unsigned)__shfl((int)base, leader, 64);
    return base + (unsigned)__popcll(vm & ((1ull << (threadIdx.x & 63u)) - 1ull));
  }
  return valid ? atomicAdd(&counters[bin], 1u) : 0u;
}

constexpr unsigned long long kCodeNone = 1ull << 62;    /* no k-mer: an unused slot of a block encodeLookupKernel reserved */
constexpr unsigned kLookupBlock = 64;                   /* slots a wave of encodeLookupKernel reserves at a time */
constexpr unsigned kShareCountStride = 64;              /* words between the shares' slot counters (a line each) */
/* The batch is cut into 8 SHARES of whole tiles, one per XCD (workgroup b works on share b % 8, the XCD it runs on under
 * round-robin dispatch: for speed only), and every bucket's place in the order into 8 sub-runs, share by share.  The
 * workgroups that append to a sub-run are then on ONE XCD: its L2 sees all the 64-byte runs that make up a line and writes
 * the line once, and the cursor it bumps is not shared with the other seven.  (One run area per bucket for the whole chip:
 * 1.2 GB written for 0.8 GB of records.) */
constexpr unsigned kShares = 8;
__host__ __device__ inline unsigned long long shareSize(unsigned long long numQueries) {
  const unsigned long long perShare = (numQueries + kShares - 1ull) / kShares;
  return (perShare + kPartitionTile - 1ull) / kPartitionTile * kPartitionTile;
}

struct BucketFormat {
  unsigned depth;      /* characters the table lookup consumes */
  unsigned bucketBits; /* min(kBucketBitsMax, 2 * depth): leading bits of the table index */
  unsigned lowBits;    /* 2 * depth - bucketBits: table-index bits below the bucket's */
  unsigned indexBits;  /* bits of a query number */
  /* round 6 (seed-bucket sharding, awfmGpuSearchOrderedRecords): the search takes the records of the buckets
   * [firstBucket, endBucket) only -- a rank's dense share of the order; endBucket 0: all of them */
  unsigned firstBucket, endBucket;
};
__host__ __device__ inline BucketFormat bucketFormat(unsigned depth, unsigned long long numQueries) {
  BucketFormat f;
  f.depth = depth;
  f.bucketBits = 2u * depth < kBucketBitsMax ? 2u * depth : kBucketBitsMax;
  f.lowBits = 2u * depth - f.bucketBits;
  f.indexBits = 1;
  while (f.indexBits < 32u && (numQueries - 1ull) >> f.indexBits) f.indexBits++;
  f.firstBucket = f.endBucket = 0u;
  return f;
}
/* does a record of a k-mer of `len` characters fit 8 bytes? */
__host__ __device__ inline bool bucketFits(unsigned len, const BucketFormat &f) {
  return len >= f.depth && 2u * len - f.bucketBits + f.indexBits <= 64u;
}
__device__ __forceinline__ unsigned bucketOf(const BucketFormat &f, unsigned long long codes) {
  return (unsigned)((codes & ((1ull << (2u * f.depth)) - 1ull)) >> f.lowBits);
}
__device__ __forceinline__ unsigned long long bucketRest(const BucketFormat &f, unsigned long long codes) {
  return ((codes >> (2u * f.depth)) << f.lowBits) | (codes & ((1ull << f.lowBits) - 1ull));
}
__device__ __forceinline__ unsigned long long bucketCodes(const BucketFormat &f, unsigned bucket, unsigned long long rest) {
  const unsigned long long low = rest & ((1ull << f.lowBits) - 1ull);
  return ((rest >> f.lowBits) << (2u * f.depth)) | ((unsigned long long)bucket << f.lowBits) | low;
}

/* pass 1: codes (unless PACKED: `chars` is the code array already) + histogram of the buckets; persistent grid, one
 * LDS histogram per workgroup, flushed once */
template <bool PACKED>
__global__ void __launch_bounds__(256)
    encodeCodesKernel(const unsigned char *__restrict__ chars, const unsigned fixedLen, const BucketFormat f,
                      const unsigned long long numQueries, unsigned long long *__restrict__ codesOut,
                      unsigned *__restrict__ hist /* [kShares][binsPad] */, const unsigned binsPad) {
  extern __shared__ unsigned sHist[]; /* 2^bucketBits + 1 */
  const unsigned bins = (1u << f.bucketBits) + 1u;
  for (unsigned e = threadIdx.x; e < bins; e += 256u) sHist[e] = 0u;
  __syncthreads();
  const unsigned share = blockIdx.x % kShares, localBlock = blockIdx.x / kShares, localGrid = gridDim.x / kShares;
  const unsigned long long size = shareSize(numQueries), first = size * share;
  const unsigned long long last = first + size < numQueries ? first + size : numQueries;
  for (unsigned long long t = first + (unsigned long long)localBlock * 256ull + threadIdx.x; t < last; t += (unsigned long long)localGrid * 256ull) {
    unsigned long long codes = 0;
    unsigned bad = 0;
    if (PACKED) {
      const unsigned long long word = ((const unsigned long long *)chars)[t];
      codes = fixedLen >= 32u ? word : (word & ((1ull << (2u * fixedLen)) - 1ull));
    } else {
      decodeKmer(chars, t * fixedLen, fixedLen, codes, bad);
      codesOut[t] = bad ? kCodeGeneral : codes;
    }
    atomicAdd(&sHist[bad ? bins - 1u : bucketOf(f, codes)], 1u);
  }
  __syncthreads();
  for (unsigned e = threadIdx.x; e < bins; e += 256u)
    if (sHist[e]) atomicAdd(&hist[share * binsPad + e], sHist[e]);
}

/* pass 1 for ASCII k-mers of a length known at compile time: a thread takes FOUR consecutive k-mers -- 4 K bytes, i.e.
 * K whole dwords wherever the batch starts -- with 16-byte loads, brings them to dword alignment once (the misalignment of
 * the batch is the same for every thread), and takes the four k-mers apart at compile-time offsets: K / 4 + 1 load
 * instructions per four k-mers instead of K / 4 + 2 per k-mer (encodeCodesKernel: one k-mer per thread at a 21-byte
 * stride), which is what bounds that kernel, not the bytes.  The last k-mers of a batch (those whose loads would run past its end) go one by one. */
typedef unsigned Dwords4 __attribute__((ext_vector_type(4), aligned(4))); /* a 16-byte load at dword alignment */
template <unsigned K>
__global__ void __launch_bounds__(256)
    encodeCodes4Kernel(const unsigned char *__restrict__ chars, const BucketFormat f, const unsigned long long numQueries,
                       unsigned long long *__restrict__ codesOut, unsigned *__restrict__ hist /* [kShares][binsPad] */,
                       const unsigned binsPad, const unsigned *__restrict__ sampleAlive = nullptr, const unsigned samples = 0u) {
  extern __shared__ unsigned sHist[]; /* 2^bucketBits + 1 */
  if (lookupChosen(sampleAlive, samples, false)) return; /* this batch is encodeLookupKernel's (uniform) */
  const unsigned bins = (1u << f.bucketBits) + 1u;
  for (unsigned e = threadIdx.x; e < bins; e += 256u) sHist[e] = 0u;
  __syncthreads();
  constexpr unsigned kLoads = (K + 1u + 3u) / 4u; /* 16-byte loads that cover K + 1 dwords */
  const unsigned shift = (unsigned)((unsigned long long)chars & 3ull);
  typedef const Dwords4 __attribute__((address_space(1))) *GlobalDwords4;
  const unsigned share = blockIdx.x % kShares, localBlock = blockIdx.x / kShares, localGrid = gridDim.x / kShares;
  const unsigned long long size = shareSize(numQueries), first = size * share; /* a multiple of the tile, hence of 4 */
  const unsigned long long last = first + size < numQueries ? first + size : numQueries;
  for (unsigned long long t = first + 4ull * ((unsigned long long)localBlock * 256ull + threadIdx.x); t < last; t += 4ull * localGrid * 256ull) {
    unsigned long long codes[4];
    unsigned bad[4];
    if (t * K + 16ull * kLoads <= numQueries * K) { /* the 16-byte loads from the aligned-down start stay inside the batch */
      const GlobalDwords4 first = (GlobalDwords4)(((unsigned long long)chars + t * K) & ~3ull);
      unsigned dw[kLoads * 4u + 1u];
#pragma unroll
      for (unsigned j = 0; j < kLoads; j++) {
        const Dwords4 q = first[j];
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
#pragma unroll
      for (unsigned i = 0; i < 4u; i++) {
        constexpr unsigned kWords = (K + 3u) / 4u; /* groups of four characters */
        const unsigned at = i * K; /* byte offset of k-mer i: compile time after unrolling */
        unsigned long long c = 0;
        unsigned b = 0;
#pragma unroll
        for (unsigned w = 0; w < 8u; w++) {
          unsigned packed = 0;
          if (w < kWords) {
            const unsigned lo = al[(at >> 2) + w], hi = (at >> 2) + w + 1u <= K ? al[(at >> 2) + w + 1u] : 0u;
            const unsigned inKmer = K - 4u * w >= 4u ? 4u : K - 4u * w; /* characters of the k-mer in this word */
            decodeWordAny(__builtin_amdgcn_alignbyte(hi, lo, at & 3u), inKmer >= 4u ? ~0u : (1u << (8u * inKmer)) - 1u, packed, b);
          }
          c = (c << 8) | packed;
        }
        codes[i] = c >> (2u * (32u - K));
        bad[i] = b;
      }
    } else {
#pragma unroll
      for (unsigned i = 0; i < 4u; i++) {
        codes[i] = 0;
        bad[i] = 0;
        if (t + i < numQueries) decodeKmer(chars, (t + i) * K, K, codes[i], bad[i]);
      }
    }
    if (t + 3ull < last) { /* the four code words are 32 consecutive bytes at a 32-byte boundary: two 16-byte stores */
      ulonglong2 *out = (ulonglong2 *)(codesOut + t);
      out[0] = make_ulonglong2(bad[0] ? kCodeGeneral : codes[0], bad[1] ? kCodeGeneral : codes[1]);
      out[1] = make_ulonglong2(bad[2] ? kCodeGeneral : codes[2], bad[3] ? kCodeGeneral : codes[3]);
    }
#pragma unroll
    for (unsigned i = 0; i < 4u; i++) {
      if (t + i < last && t + 3ull >= last) codesOut[t + i] = bad[i] ? kCodeGeneral : codes[i];
      (void)ldsCountRank(sHist, bad[i] ? bins - 1u : bucketOf(f, codes[i]), t + i < last);
    }
  }
  __syncthreads();
  for (unsigned e = threadIdx.x; e < bins; e += 256u)
    if (sHist[e]) atomicAdd(&hist[share * binsPad + e], sHist[e]);
}

/* ---- lookup first, fused: the k-mers still alive after the table are searched where they are found ----
 * encodeLookupKernel hands its survivors (4.4 % of 10^8 random 21-mers) to a scan, a partition and orderedSearchKernel, which
 * looks every survivor's table entry up again; those kernels take 0.45 ms per 10^8 k-mers and -- being chains of dependent
 * reads over a few k-mers -- hardly less per 10^7, which is what a rank of an 8-GPU run holds.  Here a wave that has looked
 * its 256 entries up puts the survivors' {codes, number, range} into LDS and takes them through their steps 16 at a time
 * (4 lanes per k-mer, the device functions of orderedSearchKernel: pair steps, flagged blocks and the odd step through the
 * one-letter image), then stores the hits: dense, or into the list.  The survivors' block reads are in input order, so
 * nothing is shared in the L2 -- but there are 1.1 * 10^7 of them against 10^8 table lines, and the 4.4 * 10^6 table
 * lines the ordered kernel read a second time are not read at all.  K-mers with ambiguity characters still go to the
 * general kernel by way of the share's region, the scan and the partition (all but empty now).
 * Results: every k-mer with hits gets the range the reference's stepping gives it (same entry, same steps); a k-mer dropped at
 * the table or emptied on the way has no hit. */
constexpr unsigned kFusedSlots = 64;    /* survivors a wave takes through the steps at a time */
constexpr unsigned kFusedCounters = 64; /* words (a line apart) the waves count their survivors into: reporting */
template <unsigned K, bool NARROW = true>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(NARROW ? 7 : 6, 8)))
    lookupSearchKernel(const DevIndex ix, const unsigned char *__restrict__ chars, const BucketFormat f, const unsigned useNext,
                       const unsigned long long numQueries, unsigned long long *__restrict__ codesOut,
                       unsigned *__restrict__ numbersOut, unsigned *__restrict__ shareCount, unsigned *__restrict__ hist,
                       const unsigned binsPad, const unsigned *__restrict__ sampleAlive, const unsigned samples,
                       ulonglong2 *__restrict__ ranges, unsigned *__restrict__ counts, const SparseOut sparse,
                       unsigned *__restrict__ keptCounters) {
  constexpr int G = 4;
  /* NARROW: 32-bit positions (awfmImageNarrow); otherwise (round 6) the 64-bit arithmetic of ref src/AwFmIndex.h:88-91 -- the
   * entries' format (DevIndex::deepNarrow: 1 or 2) is read at run time either way */
  typedef typename PositionType<NARROW>::type pos_t;
  /* kernel arguments, uniform (one instantiation per k-mer length, not four): pair steps when the image has its pair
   * blocks; the hits into the list when there is one */
  const bool PAIR = ix.pairBlocks != nullptr && (useNext & 2u) == 0u;
  const bool LIST = sparse.count != nullptr;
  __shared__ unsigned long long sC[24];
  __shared__ unsigned sMask[(kBlockMask + 1) * kSlices];
  __shared__ unsigned long long sSuper[NARROW ? 1 : kMaxNucSuper * 4];
  __shared__ unsigned long long sPairC[16];
  extern __shared__ unsigned sPairSuper[];
  __shared__ unsigned long long sCodes[4][kFusedSlots];
  __shared__ unsigned sNum[4][kFusedSlots];
  __shared__ pos_t sSp[4][kFusedSlots], sEp[4][kFusedSlots];
  constexpr unsigned kHitBuffer = 32;
  __shared__ unsigned sHitKmers[4][kHitBuffer];
  __shared__ unsigned long long sHitRanges[4][kHitBuffer][2];
  __shared__ unsigned sHitLeft[4], sWavesDone;
  if (!lookupChosen(sampleAlive, samples, true)) return; /* this batch is encodeCodes4Kernel's (uniform) */
  if (threadIdx.x == 0) sWavesDone = 0u;
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  stageMaskTable(sMask);
  nucStageSuper<NARROW>(ix, sSuper);
  if (PAIR) pairStageTables<NARROW, 16u>(ix, sPairC, sPairSuper);
  __syncthreads();
  const unsigned bins = (1u << f.bucketBits) + 1u;
  constexpr unsigned kLoads = (K + 1u + 3u) / 4u;
  const unsigned shift = (unsigned)((unsigned long long)chars & 3ull);
  typedef const Dwords4 __attribute__((address_space(1))) *GlobalDwords4;
  const unsigned share = blockIdx.x % kShares, localBlock = blockIdx.x / kShares, localGrid = gridDim.x / kShares;
  const unsigned long long size = shareSize(numQueries), first = size * share;
  const unsigned long long last = first + size < numQueries ? first + size : numQueries;
  const unsigned long long tableMask = (1ull << (2u * f.depth)) - 1ull;
  const unsigned lengthBits = deepLengthBits(ix);
  const unsigned lane = threadIdx.x & 63u, gl = threadIdx.x % G, firstSlice = gl;
  const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  unsigned blockBase = 0, blockUsed = 0, blockSlots = 0; /* the wave's block of slots in its share's region: general k-mers */
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
  /* (Round 5 tried dealing the 256-k-mer trips to the waves from a counter per share, the next ticket drawn a trip ahead, so
   * that the waves with long trips do not keep the chip waiting at the end: 0.39 instead of 0.36 ms per 1.25 * 10^7 k-mers, 2.89
   * instead of 2.68 per 10^8, and an 81st register: the fixed stride stays, with the grid trimmed to even trips.)
   * Then the waves' own clocks were looked at (round 5's timeline build: profiles/r5/lookup_timeline): every wave makes the
   * same number of trips, and the waves of a launch end anywhere between 0.45 and 1.0 of its span, in the order of their SLOT
   * on the SIMD (slot 0: 23-25 us a trip, slot 6: 48) -- whatever serves the waves' memory requests serves the older slots
   * first, and rotating the issue priority (s_setprio by trip and slot) changes nothing about it.  Dealing the trips out again,
   * without a register this time -- the first 0 / 25 / 50 / 75 % of a wave's trips at the fixed stride, the rest of the share
   * by tickets drawn when needed, with and without taking tickets of the other XCDs' shares once the own one is dealt out --
   * lets every wave run to the end (slot 0 then makes 110-130 trips, slot 6 two to five, of 300-1500 us each) and is slower
   * however it is mixed: 2.85-2.96 against 2.70-2.80 ms per 10^8, 0.47-0.49 against 0.36-0.38 per 1.25 * 10^7 (the last
   * trips of the starved waves are the tail; keeping those waves from drawing late tickets by looking at the counter first
   * costs more than all of it: 6 ms).  With equal shares the early slots finish, the later ones move up, and the launch as a
   * whole runs at the memory system's rate: the spread of the ends is how the hardware shares, not time that is lost. */
  const unsigned long long waveFirst = first + 4ull * ((unsigned long long)localBlock * 256ull + (threadIdx.x & ~63u));
  for (unsigned long long tw = waveFirst; tw < last; tw += 4ull * localGrid * 256ull) {
    const unsigned long long t = tw + 4ull * lane;
    unsigned long long codes[4];
    unsigned bad[4];
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
      unsigned al[K + 1u];
#pragma unroll
      for (unsigned j = 0; j < K; j++) al[j] = __builtin_amdgcn_alignbyte(dw[j + 1u], dw[j], shift);
      al[K] = 0u;
#pragma unroll
      for (unsigned i = 0; i < 4u; i++) {
        constexpr unsigned kWords = (K + 3u) / 4u;
        const unsigned at = i * K;
        unsigned long long c = 0;
        unsigned b = 0;
#pragma unroll
        for (unsigned x = 0; x < 8u; x++) {
          unsigned packed = 0;
          if (x < kWords) {
            const unsigned lo = al[(at >> 2) + x], hi = (at >> 2) + x + 1u <= K ? al[(at >> 2) + x + 1u] : 0u;
            const unsigned inKmer = K - 4u * x >= 4u ? 4u : K - 4u * x;
            decodeWordAny(__builtin_amdgcn_alignbyte(hi, lo, at & 3u), inKmer >= 4u ? ~0u : (1u << (8u * inKmer)) - 1u, packed, b);
          }
          c = (c << 8) | packed;
        }
        codes[i] = c >> (2u * (32u - K));
        bad[i] = b;
      }
    } else {
#pragma unroll
      for (unsigned i = 0; i < 4u; i++) {
        codes[i] = 0;
        bad[i] = 0;
        if (t + i < numQueries) decodeKmer(chars, (t + i) * K, K, codes[i], bad[i]);
      }
    }
    uint2 entry[4];
#pragma unroll
    for (unsigned i = 0; i < 4u; i++) entry[i] = ((const uint2 *)ix.deepSeed)[t + i < last && !bad[i] ? (codes[i] & tableMask) : 0ull];
    /* a survivor is searched here when it is among the first kFusedSlots of its round (a round of 256 random 21-mers has
     * 11); the others -- and the k-mers with ambiguity characters, which are the general kernel's -- go to the share's
     * region as encodeLookupKernel's survivors do, and the kernels behind this one take them from there.  Nothing of a
     * k-mer is kept in registers past this point: the steps below work out of LDS. */
    unsigned long long amask[4];
    unsigned abefore[4], atotal = 0, stotal = 0;
    bool append[4];
#pragma unroll
    for (unsigned i = 0; i < 4u; i++) {
      const unsigned length = entry[i].y & lengthBits;
      const bool bit = (useNext & 1u) == 0u || ((entry[i].y >> (16u + ((unsigned)(codes[i] >> (2u * f.depth)) & 15u))) & 1u) != 0u;
      const bool general = t + i < last && bad[i] != 0u;
      const bool survives = t + i < last && bad[i] == 0u && length != 0u && bit;
      const unsigned long long smask = __ballot(survives);
      const unsigned rank = stotal + (unsigned)__popcll(smask & ((1ull << lane) - 1ull));
      stotal += (unsigned)__popcll(smask);
      if (survives && rank < kFusedSlots) {
        const ulonglong2 r = deepSeedOpen(ix, codes[i] & tableMask, entry[i], nullptr);
        sCodes[w][rank] = codes[i];
        sNum[w][rank] = (unsigned)(t + i);
        sSp[w][rank] = (pos_t)r.x;
        sEp[w][rank] = (pos_t)r.y;
      }
      append[i] = general || (survives && rank >= kFusedSlots);
      amask[i] = __ballot(append[i]);
      abefore[i] = atotal;
      atotal += (unsigned)__popcll(amask[i]);
    }
    if (atotal != 0u && (useNext & 4u) != 0u) { /* wave-uniform; rare */
      /* LOOKUP ONLY (the search launched no ordered kernels behind this one: the sample of an earlier batch of the stream
       * said this kernel would be the one, awfm_gpu_ordered.hip): what is not searched here is the general kernel's, which
       * takes the LAST *shareCount 8-byte records of numbersOut[0 .. numQueries) -- a k-mer number each */
      unsigned base = 0;
      if (lane == 0) base = atomicAdd(shareCount, atotal);
      base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
      for (unsigned i = 0; i < 4u; i++)
        if (append[i]) {
          const unsigned long long rec = numQueries - 1ull - (base + abefore[i] + (unsigned)__popcll(amask[i] & ((1ull << lane) - 1ull)));
          ((unsigned long long *)numbersOut)[rec] = (unsigned long long)(t + i);
        }
    } else if (atotal != 0u) { /* wave-uniform; rare */
      if (blockUsed + atotal > blockSlots) {
        if (blockUsed + lane < blockSlots) codesOut[first + blockBase + blockUsed + lane] = kCodeNone;
        blockSlots = atotal > kLookupBlock || tw + 256ull > last ? atotal : kLookupBlock;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&shareCount[share * kShareCountStride], blockSlots);
        blockBase = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        blockUsed = 0;
      }
#pragma unroll
      for (unsigned i = 0; i < 4u; i++)
        if (append[i]) {
          const unsigned long long slot = first + blockBase + blockUsed + abefore[i] + (unsigned)__popcll(amask[i] & ((1ull << lane) - 1ull));
          codesOut[slot] = bad[i] ? kCodeGeneral : codes[i];
          numbersOut[slot] = (unsigned)(t + i);
          atomicAdd(&hist[share * binsPad + (bad[i] ? bins - 1u : bucketOf(f, codes[i]))], 1u);
        }
      blockUsed += atotal;
    }
    const unsigned inBatch = stotal < kFusedSlots ? stotal : kFusedSlots;
    keptHere += inBatch;
    if (inBatch != 0u) { /* wave-uniform */
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (unsigned pass = 0; pass < inBatch; pass += 64u / G) { /* wave-uniform */
        const unsigned slot = pass + lane / G;
        const bool live = slot < inBatch;
        pos_t sp = 1, ep = 0;
        unsigned long long rem = 0;
        unsigned index = 0;
        int pos = -1;
        if (live) {
          const unsigned long long c = sCodes[w][slot];
          index = sNum[w][slot];
          sp = sSp[w][slot];
          ep = sEp[w][slot];
          rem = c >> (2u * f.depth);
          pos = (int)(K - f.depth) - 1;
        }
        if (PAIR) {
          while (pos >= 1 && sp <= ep) {
            const unsigned c2 = (unsigned)rem & 3u, c1 = (unsigned)(rem >> 2) & 3u;
            if (pairSearchStep<NARROW>(ix, sPairC, sPairSuper, sMask, gl, c1 * 4u + c2, sp, ep) == kPairFlagged) {
              nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, c2, sp, ep);
              if (sp <= ep) nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, c1, sp, ep);
            }
            pos -= 2;
            rem >>= 4;
          }
          if (pos == 0 && sp <= ep) {
            nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, (unsigned)rem & 3u, sp, ep);
            pos--;
          }
        } else {
          while (pos >= 0 && sp <= ep) {
            nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, (unsigned)rem & 3u, sp, ep);
            pos--;
            rem >>= 2;
          }
        }
        const bool hit = live && gl == 0 && sp <= ep;
        if (LIST) {
          const unsigned long long hitMask = __ballot(hit);
          if (hitMask != 0ull) { /* wave-uniform; at most 16 hits a pass */
            const unsigned hits = (unsigned)__builtin_amdgcn_readfirstlane((int)__popcll(hitMask));
            if (hit) {
              const unsigned at = hitFill + (unsigned)__popcll(hitMask & ((1ull << lane) - 1ull));
              sHitKmers[w][at] = index;
              sHitRanges[w][at][0] = (unsigned long long)sp;
              sHitRanges[w][at][1] = (unsigned long long)ep;
            }
            hitFill = (unsigned)__builtin_amdgcn_readfirstlane((int)(hitFill + hits));
          }
          if (hitFill + 64u / G > kHitBuffer) flushHits();
        } else if (hit) {
          if (ranges) ranges[index] = make_ulonglong2((unsigned long long)sp, (unsigned long long)ep);
          if (counts) counts[index] = (unsigned)(ep - sp + (pos_t)1);
        }
      }
      __builtin_amdgcn_wave_barrier(); /* the slots are written again by the next round */
    }
  }
  for (unsigned at = blockUsed + lane; at < blockSlots; at += 64u) codesOut[first + blockBase + at] = kCodeNone;
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

/* Everything a bucketed search needs before its front end, in ONE launch (round 5; they were a memset, a fill kernel and
 * sampleAliveKernel: 17 of the 470 us a 1.25 * 10^7-k-mer shard takes): the scratch counters zeroed (two regions of 16-byte
 * pieces), the caller's list of hits pre-filled, and the sample taken.  The sample's word is not zeroed by the launch that
 * adds to it -- the workgroups of a launch are not ordered -- : a search leaves the OTHER of two words zero for the search
 * that uses the scratch slot next (`aliveNext`; the host keeps the parity).  The word is 64 bits: every sampling workgroup
 * adds {1, its count} in one atomic, so the one whose add returns "all the others are in" knows the total and publishes
 * it -- {number of this search, k-mers alive} -- in page-locked host memory, where a later search of the stream reads it
 * without a wait and launches only the front end the sample chose (awfm_gpu_ordered.hip: lookup prediction).  The front
 * ends read the low half of the word as sampleAliveKernel's count. */
template <bool AMINO>
__global__ void __launch_bounds__(256)
    lookupPrepKernel(const DevIndex ix, const unsigned char *__restrict__ chars, const unsigned fixedLen, const unsigned depth,
                     const unsigned useNext, const unsigned long long numQueries, const unsigned samples,
                     unsigned long long *__restrict__ aliveOut, unsigned long long *__restrict__ aliveNext,
                     uint4 *__restrict__ zeroA, const unsigned vecsA, uint4 *__restrict__ zeroB, const unsigned vecsB,
                     const SparseOut fill, unsigned long long *__restrict__ verdictHost, const unsigned searchNumber) {
  const unsigned gid = blockIdx.x * 256u + threadIdx.x, gsize = gridDim.x * 256u;
  for (unsigned i = gid; i < vecsA; i += gsize) zeroA[i] = make_uint4(0u, 0u, 0u, 0u);
  for (unsigned i = gid; i < vecsB; i += gsize) zeroB[i] = make_uint4(0u, 0u, 0u, 0u);
  if (fill.count) { /* the list of hits (awfmGpuSearchHitsCompact): empty entries sort behind every k-mer */
    for (unsigned i = gid; i < fill.cap; i += gsize) {
      fill.kmers[i] = 0xFFFFFFFFu;
      fill.ranges[i] = make_ulonglong2(1ull, 0ull);
    }
    if (gid == 0u) *fill.count = 0u;
  }
  if (gid == 0u) *aliveNext = 0ull;
  const unsigned sampleBlocks = (samples + 255u) / 256u; /* uniform per workgroup from here */
  if (blockIdx.x >= sampleBlocks) return;
  __shared__ AminoShared sAmino;
  __shared__ unsigned sAlive;
  if (AMINO) aminoStageTables(sAmino);
  if (threadIdx.x == 0) sAlive = 0u;
  __syncthreads();
  bool alive = false;
  if (gid < samples) {
    const unsigned long long t = (unsigned long long)gid * (numQueries / samples);
    if (AMINO) { /* aminoLookupSearchKernel's test (awfm_amino_lookup_kernel.h): the entry over the last `depth` characters */
      const unsigned char *at = chars + t * fixedLen;
      unsigned idx = 0;
      bool bad = false;
      for (unsigned c = fixedLen - depth; c < fixedLen; c++) {
        const unsigned letter = aminoLetterIndex(sAmino, at[c]);
        idx = idx * 20u + letter;
        bad |= letter >= 20u;
      }
      const uint2 e = ((const uint2 *)ix.deepSeed)[bad ? 0u : idx];
      const unsigned nextLetter = fixedLen > depth ? aminoLetterIndex(sAmino, at[fixedLen - depth - 1u]) : 20u;
      alive = bad || (aminoDeepLength(ix, e) != 0u && aminoDeepNextBit(ix, e, nextLetter));
    } else {
      unsigned long long codes = 0;
      unsigned bad = 0;
      decodeKmer(chars, t * fixedLen, fixedLen, codes, bad);
      if (bad) {
        alive = true;
      } else {
        const uint2 e = ((const uint2 *)ix.deepSeed)[codes & ((1ull << (2u * depth)) - 1ull)];
        const unsigned length = e.y & deepLengthBits(ix);
        alive = length != 0u && (!useNext || ((e.y >> (16u + ((unsigned)(codes >> (2u * depth)) & 15u))) & 1u) != 0u);
      }
    }
  }
  const unsigned n = (unsigned)__popcll(__ballot(alive));
  if ((threadIdx.x & 63u) == 0 && n) atomicAdd(&sAlive, n);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long old = atomicAdd(aliveOut, (1ull << 32) | (unsigned long long)sAlive);
    if ((unsigned)(old >> 32) == sampleBlocks - 1u && verdictHost)
      *(volatile unsigned long long *)verdictHost = ((unsigned long long)searchNumber << 32) | (unsigned long long)((unsigned)old + sAlive);
  }
}

/* The same scan for the shared histogram hist[kShares][binsPad]: bucketStart as below (a bucket's sub-runs are contiguous,
 * share by share), and hist[share][b] is overwritten with where the share's sub-run of bucket b begins */
__global__ void __launch_bounds__(1024)
    bucketScanSharesKernel(unsigned *__restrict__ hist, const unsigned bins, const unsigned binsPad, unsigned *__restrict__ bucketStart,
                           unsigned *__restrict__ generalCount, const unsigned generalEnd /* records in the array: the batch's k-mers */) {
  __shared__ unsigned sWave[16];
  constexpr unsigned kPer = 3; /* 3 x 1024 >= 2049 */
  unsigned v[kPer], sum = 0;
  for (unsigned j = 0; j < kPer; j++) {
    const unsigned e = threadIdx.x * kPer + j;
    v[j] = 0u;
    if (e < bins)
      for (unsigned sh = 0; sh < kShares; sh++) v[j] += hist[sh * binsPad + e];
    sum += v[j];
  }
  unsigned incl = sum;
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned up = __shfl_up(incl, off);
    if ((int)(threadIdx.x & 63u) >= off) incl += up;
  }
  if ((threadIdx.x & 63u) == 63u) sWave[threadIdx.x >> 6] = incl;
  __syncthreads();
  unsigned before = 0;
  for (unsigned w = 0; w < (threadIdx.x >> 6); w++) before += sWave[w];
  unsigned running = before + incl - sum;
  for (unsigned j = 0; j < kPer; j++) {
    const unsigned e = threadIdx.x * kPer + j;
    if (e <= bins) bucketStart[e] = running; /* entry `bins` = the total */
    if (e < bins) {
      /* the last bin (k-mers left to the general kernel) sits at the END of the record array, where that kernel looks for
       * it -- the same place as `running` when every k-mer of the batch is in some bin, behind a gap after encodeLookupKernel */
      unsigned at = e == bins - 1u ? generalEnd - v[j] : running;
      for (unsigned sh = 0; sh < kShares; sh++) {
        const unsigned n = hist[sh * binsPad + e];
        hist[sh * binsPad + e] = at;
        at += n;
      }
      if (e == bins - 1u) *generalCount = v[j];
    }
    running += v[j];
  }
}

/* bucketStart[b] = k-mers in the buckets before b (2^bucketBits + 2 entries: [2^bucketBits] = k-mers the ordered
 * kernel covers, [2^bucketBits + 1] = all); generalCount = size of the last bin; one workgroup */
__global__ void __launch_bounds__(1024)
    bucketScanKernel(const unsigned *__restrict__ hist, const unsigned bins, unsigned *__restrict__ bucketStart,
                     unsigned *__restrict__ generalCount) {
  __shared__ unsigned sWave[16];
  constexpr unsigned kPer = 3; /* 3 x 1024 >= 2049 */
  unsigned v[kPer], sum = 0;
  for (unsigned j = 0; j < kPer; j++) {
    const unsigned e = threadIdx.x * kPer + j;
    v[j] = e < bins ? hist[e] : 0u;
    sum += v[j];
  }
  unsigned incl = sum;
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned up = __shfl_up(incl, off);
    if ((int)(threadIdx.x & 63u) >= off) incl += up;
  }
  if ((threadIdx.x & 63u) == 63u) sWave[threadIdx.x >> 6] = incl;
  __syncthreads();
  unsigned before = 0;
  for (unsigned w = 0; w < (threadIdx.x >> 6); w++) before += sWave[w];
  unsigned running = before + incl - sum;
  for (unsigned j = 0; j < kPer; j++) {
    const unsigned e = threadIdx.x * kPer + j;
    if (e <= bins) bucketStart[e] = running; /* entry `bins` = the total */
    running += v[j];
  }
  if (threadIdx.x == 0) *generalCount = hist[bins - 1u];
}

/* pass 2: codes -> records, partitioned by bucket.  subStart[share][b] = where the share's sub-run of bucket b begins,
 * cursors[share][b] = records placed in it so far. */
__global__ void __launch_bounds__(kPartitionThreads)
    partitionKernel(const unsigned long long *__restrict__ codes, const unsigned fixedLen, const BucketFormat f,
                    const unsigned long long numQueries, const unsigned *__restrict__ subStart,
                    unsigned *__restrict__ cursors, unsigned long long *__restrict__ recs, const unsigned honourGeneral,
                    const unsigned *__restrict__ shareCountIn = nullptr /* after encodeLookupKernel: code words in the share's region */,
                    const unsigned *__restrict__ numbersIn = nullptr /* ... and the k-mer numbers beside them */,
                    const unsigned *__restrict__ sampleAlive = nullptr, const unsigned samples = 0u /* lookupChosen: was it that kernel? */,
                    const unsigned long long numberBase = 0ull /* a shard of a larger batch (awfmGpuOrderKmers): its first k-mer's number there */) {
  const bool afterLookup = shareCountIn != nullptr && lookupChosen(sampleAlive, samples, true);
  const unsigned *__restrict__ shareCount = afterLookup ? shareCountIn : nullptr;
  const unsigned *__restrict__ numbers = afterLookup ? numbersIn : nullptr;
  extern __shared__ unsigned long long sDyn[];
  unsigned long long *sRec = sDyn;                       /* kPartitionTile records, bucket by bucket */
  unsigned *sCnt = (unsigned *)(sRec + kPartitionTile);  /* records of the tile per bucket */
  const unsigned bins = (1u << f.bucketBits) + 1u;
  const unsigned binsPad = (bins + 3u) & ~3u;
  unsigned *sLoc = sCnt + binsPad;                       /* where a bucket's run starts in sRec */
  unsigned *sDst = sLoc + binsPad;                       /* where it goes in recs */
  __shared__ unsigned sWave[kPartitionThreads / 64];
  constexpr unsigned kLongRun = 256;
  __shared__ unsigned sLong[kPartitionTile / kLongRun], sNumLong;
  if (threadIdx.x == 0) sNumLong = 0u;
  const unsigned long long lenMask = fixedLen >= 32u ? ~0ull : ((1ull << (2u * fixedLen)) - 1ull);
  /* the tiles of this workgroup's share */
  const unsigned share = blockIdx.x % kShares, localBlock = blockIdx.x / kShares, localGrid = gridDim.x / kShares;
  const unsigned long long tilesPerShare = shareSize(numQueries) / kPartitionTile;
  const unsigned long long shareFirst = shareSize(numQueries) * share;
  const unsigned long long shareEnd = shareCount ? shareFirst + shareCount[share * kShareCountStride]
                                                 : (shareFirst + shareSize(numQueries) < numQueries ? shareFirst + shareSize(numQueries) : numQueries);
  const unsigned long long tiles = tilesPerShare * share + (shareEnd > shareFirst ? (shareEnd - shareFirst + kPartitionTile - 1ull) / kPartitionTile : 0ull);
  const unsigned *myStart = subStart + share * binsPad;
  unsigned *myCursors = cursors + share * binsPad;
  constexpr unsigned kPer = 3; /* bins handled per thread in the scan: 3 x 1024 >= 2049 */
  /* two register sets: the codes of the tile after the current one are requested before the current one is touched, so
   * that their way from memory overlaps all of its phases (one workgroup per CU: nothing else hides it).  What bounds
   * the kernel are its stores -- 2048 runs of 64 bytes on average per tile, 1.2 GB written for 0.8 GB of records: with
   * every tile storing to the same place (wrong results, measurement only) it takes 0.34 instead of 0.63 ms; with
   * per-tile offsets from a scan instead of the atomic reservations (a stable partition, three small launches more)
   * 0.59 + 0.12 ms. */
  unsigned long long recA[kPartitionItems], recB[kPartitionItems];
  auto loadTile = [&](unsigned long long tile, unsigned long long *rec) {
    const unsigned long long tileBase = tile * kPartitionTile;
#pragma unroll
    for (unsigned j = 0; j < kPartitionItems; j++) {
      const unsigned long long idx = tileBase + (unsigned long long)j * kPartitionThreads + threadIdx.x;
      rec[j] = idx < shareEnd ? codes[idx] : 0ull;
    }
  };
  auto processTile = [&](unsigned long long tile, unsigned long long *rec) {
    for (unsigned e = threadIdx.x; e < bins; e += kPartitionThreads) sCnt[e] = 0u;
    __syncthreads();
    const unsigned long long tileBase = tile * kPartitionTile;
    unsigned where[kPartitionItems]; /* bucket << 16 | rank inside the tile's run (a tile has 16384 records) */
#pragma unroll
    for (unsigned j = 0; j < kPartitionItems; j++) {
      const unsigned long long idx = tileBase + (unsigned long long)j * kPartitionThreads + threadIdx.x;
      where[j] = 0xFFFFFFFFu;
      const bool valid = idx < shareEnd && !(shareCount && rec[j] == kCodeNone); /* (kCodeNone: an unused slot of encodeLookupKernel's) */
      const bool general = honourGeneral != 0u && (rec[j] & kCodeGeneral) != 0ull; /* never for bit-packed input: every word is a k-mer */
      const unsigned long long c = rec[j] & lenMask;
      const unsigned b = general ? bins - 1u : bucketOf(f, c);
      const unsigned rank = ldsCountRank(sCnt, b, valid);
      if (valid) {
        where[j] = (b << 16) | rank;
        rec[j] = (general ? 0ull : bucketRest(f, c) << f.indexBits) | ((numbers ? (unsigned long long)numbers[idx] : idx) + numberBase);
      }
    }
    __syncthreads();
    { /* exclusive scan of the counts; one global reservation per non-empty bucket, all of a thread's in flight together */
      unsigned v[kPer], got[kPer], start[kPer], sum = 0;
#pragma unroll
      for (unsigned j = 0; j < kPer; j++) {
        const unsigned e = threadIdx.x * kPer + j;
        v[j] = e < bins ? sCnt[e] : 0u;
        sum += v[j];
      }
#pragma unroll
      for (unsigned j = 0; j < kPer; j++) {
        const unsigned e = threadIdx.x * kPer + j;
        got[j] = v[j] ? atomicAdd(&myCursors[e], v[j]) : 0u;
        start[j] = v[j] ? myStart[e] : 0u;
      }
      unsigned incl = sum;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned up = __shfl_up(incl, off);
        if ((int)(threadIdx.x & 63u) >= off) incl += up;
      }
      if ((threadIdx.x & 63u) == 63u) sWave[threadIdx.x >> 6] = incl;
      __syncthreads();
      unsigned running = incl - sum;
      for (unsigned w = 0; w < (threadIdx.x >> 6); w++) running += sWave[w];
#pragma unroll
      for (unsigned j = 0; j < kPer; j++) {
        const unsigned e = threadIdx.x * kPer + j;
        if (e < bins) {
          sLoc[e] = running;
          sDst[e] = start[j] + got[j];
        }
        running += v[j];
      }
    }
    __syncthreads();
#pragma unroll
    for (unsigned j = 0; j < kPartitionItems; j++)
      if (where[j] != 0xFFFFFFFFu) sRec[sLoc[where[j] >> 16] + (where[j] & 0xFFFFu)] = rec[j];
    __syncthreads();
    /* runs out: 8 lanes per bucket, 8 buckets per wave instruction; a run of more than kLongRun records (a sorted batch:
     * the whole tile is one run) is left to the whole workgroup */
    for (unsigned b = (threadIdx.x >> 3); b < bins; b += kPartitionThreads / 8u) {
      const unsigned count = sCnt[b], loc = sLoc[b];
      if (count != 0u && count <= kLongRun) {
        const unsigned long long dst = sDst[b];
        for (unsigned j = threadIdx.x & 7u; j < count; j += 8u) recs[dst + j] = sRec[loc + j];
      } else if (count > kLongRun && (threadIdx.x & 7u) == 0u) {
        sLong[atomicAdd(&sNumLong, 1u)] = b; /* at most kPartitionTile / kLongRun of them */
      }
    }
    __syncthreads();
    const unsigned numLong = sNumLong;
    for (unsigned k = 0; k < numLong; k++) {
      const unsigned b = sLong[k], count = sCnt[b], loc = sLoc[b];
      const unsigned long long dst = sDst[b];
      for (unsigned j = threadIdx.x; j < count; j += kPartitionThreads) recs[dst + j] = sRec[loc + j];
    }
    __syncthreads();
    if (threadIdx.x == 0) sNumLong = 0u;
  };
  unsigned long long tile = tilesPerShare * share + localBlock;
  if (tile < tiles) loadTile(tile, recA);
  while (tile < tiles) {
    if (tile + localGrid < tiles) loadTile(tile + localGrid, recB);
    processTile(tile, recA);
    tile += localGrid;
    if (tile >= tiles) break;
    if (tile + localGrid < tiles) loadTile(tile + localGrid, recA);
    processTile(tile, recB);
    tile += localGrid;
  }
}

/*
 * The same two passes for batches whose records are 16 bytes (QueryRec: mixed-length batches, fixed-length k-mers of more
 * than 24 characters): encodeRecordsKernel writes the records in batch order and counts buckets, partitionRecordsKernel
 * puts them in bucket order (tiles of 8192 records).  A record carries its whole code string and its length, so the search
 * kernel needs nothing but the order.  Replaces encode + two radix-sort passes over (key, 16-byte record) pairs:
 * 75 instead of 117 bytes of memory traffic per mixed-length k-mer.
 */
constexpr unsigned kWideItems = 8, kWideTile = kPartitionThreads * kWideItems;
constexpr unsigned kWideBucketShift = 15u - kBucketBitsMax; /* bucket = 15-bit start key >> 4 */

__device__ __forceinline__ unsigned wideBucket(const QueryRec &r, unsigned seedK, unsigned deepK, unsigned fixedDepth, bool varlen) {
  if (r.length == 0xFFFFFFFFu) return 1u << kBucketBitsMax; /* left to the general kernel: the last bin */
  const unsigned depth = varlen ? orderStartDepth(r.length, seedK, deepK) : fixedDepth;
  unsigned key = startKey15(r.codes, r.length, depth) >> kWideBucketShift;
  if (varlen) {
    /* the k-mers of a wave round finish together when they have about as many memory rounds to go (pair steps from a
     * table, single steps from a letter range): the two low bits of the bucket say how many, 9 bits are left for the place
     * in the BWT.  10^8 mixed 8..30-mers: search kernel 8.7-8.8 -> 7.4-7.8 ms (three class bits: 7.7; the class as the
     * MAJOR key: 10.2) */
    const unsigned rounds = depth != 0u ? (r.length - depth + 1u) / 2u : r.length - 1u;
    const unsigned cls = rounds <= 1u ? 0u : (rounds <= 3u ? 1u : (rounds <= 5u ? 2u : 3u));
    key = (key & ~3u) | cls;
  }
  return key;
}

template <bool VARLEN>
__global__ void __launch_bounds__(256)
    encodeRecordsKernel(const unsigned char *__restrict__ chars, const unsigned long long *__restrict__ offsets, const unsigned fixedLen,
                        const unsigned fixedDepth, const unsigned seedK, const unsigned deepK, const unsigned long long numQueries,
                        QueryRec *__restrict__ recs, unsigned *__restrict__ hist, const unsigned *__restrict__ sampleAlive = nullptr,
                        const unsigned samples = 0u) {
  __shared__ unsigned sHist[(1u << kBucketBitsMax) + 1u];
  constexpr unsigned bins = (1u << kBucketBitsMax) + 1u;
  if (lookupChosen(sampleAlive, samples, false)) return; /* this batch is mixedLookupSearchKernel's (uniform) */
  for (unsigned e = threadIdx.x; e < bins; e += 256u) sHist[e] = 0u;
  __syncthreads();
  const unsigned long long tiles = (numQueries + 255ull) / 256ull;
  for (unsigned long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const unsigned long long t = tile * 256ull + threadIdx.x;
    if (t >= numQueries) continue;
    unsigned long long start, codes = 0;
    unsigned len = fixedLen, bad = 0;
    if (VARLEN) {
      start = offsets[t];
      const unsigned long long l = offsets[t + 1] - start;
      len = l > 33ull ? 33u : (unsigned)l;
    } else {
      start = t * fixedLen;
    }
    const bool inRange = len >= 1u && len <= 32u;
    if (inRange) decodeKmer(chars, start, len, codes, bad);
    QueryRec r;
    r.codes = codes;
    r.index = (unsigned)t;
    r.length = inRange && bad == 0u ? len : 0xFFFFFFFFu;
    recs[t] = r;
    (void)ldsCountRank(sHist, wideBucket(r, seedK, deepK, fixedDepth, VARLEN), true);
  }
  __syncthreads();
  for (unsigned e = threadIdx.x; e < bins; e += 256u)
    if (sHist[e]) atomicAdd(&hist[e], sHist[e]);
}

template <bool VARLEN>
__global__ void __launch_bounds__(kPartitionThreads)
    partitionRecordsKernel(const QueryRec *__restrict__ in, const unsigned fixedDepth, const unsigned seedK, const unsigned deepK,
                           const unsigned long long numQueries, const unsigned *__restrict__ bucketStart,
                           unsigned *__restrict__ cursors, QueryRec *__restrict__ out, const unsigned *__restrict__ sampleAlive = nullptr,
                           const unsigned samples = 0u) {
  extern __shared__ unsigned long long sDyn[];
  if (lookupChosen(sampleAlive, samples, false)) return; /* this batch is mixedLookupSearchKernel's (uniform) */
  ulonglong2 *sRec = (ulonglong2 *)sDyn;            /* kWideTile records, bucket by bucket */
  unsigned *sCnt = (unsigned *)(sRec + kWideTile);
  constexpr unsigned bins = (1u << kBucketBitsMax) + 1u, binsPad = (bins + 3u) & ~3u;
  unsigned *sLoc = sCnt + binsPad, *sDst = sLoc + binsPad;
  __shared__ unsigned sWave[kPartitionThreads / 64];
  constexpr unsigned kLongRun = 256;
  __shared__ unsigned sLong[kWideTile / kLongRun], sNumLong;
  if (threadIdx.x == 0) sNumLong = 0u;
  const unsigned long long tiles = (numQueries + kWideTile - 1ull) / kWideTile;
  constexpr unsigned kPer = 3;
  for (unsigned long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    for (unsigned e = threadIdx.x; e < bins; e += kPartitionThreads) sCnt[e] = 0u;
    __syncthreads();
    const unsigned long long tileBase = tile * kWideTile;
    ulonglong2 rec[kWideItems];
    unsigned where[kWideItems];
#pragma unroll
    for (unsigned j = 0; j < kWideItems; j++) {
      const unsigned long long idx = tileBase + (unsigned long long)j * kPartitionThreads + threadIdx.x;
      rec[j] = idx < numQueries ? *(const ulonglong2 *)(in + idx) : make_ulonglong2(0ull, 0ull);
    }
#pragma unroll
    for (unsigned j = 0; j < kWideItems; j++) {
      const unsigned long long idx = tileBase + (unsigned long long)j * kPartitionThreads + threadIdx.x;
      where[j] = 0xFFFFFFFFu;
      QueryRec r;
      r.codes = rec[j].x;
      r.index = (unsigned)rec[j].y;
      r.length = (unsigned)(rec[j].y >> 32);
      const unsigned b = wideBucket(r, seedK, deepK, fixedDepth, VARLEN);
      const unsigned rank = ldsCountRank(sCnt, b, idx < numQueries);
      if (idx < numQueries) where[j] = (b << 16) | rank;
    }
    __syncthreads();
    {
      unsigned v[kPer], got[kPer], start[kPer], sum = 0;
#pragma unroll
      for (unsigned j = 0; j < kPer; j++) {
        const unsigned e = threadIdx.x * kPer + j;
        v[j] = e < bins ? sCnt[e] : 0u;
        sum += v[j];
      }
#pragma unroll
      for (unsigned j = 0; j < kPer; j++) {
        const unsigned e = threadIdx.x * kPer + j;
        got[j] = v[j] ? atomicAdd(&cursors[e], v[j]) : 0u;
        start[j] = v[j] ? bucketStart[e] : 0u;
      }
      unsigned incl = sum;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned up = __shfl_up(incl, off);
        if ((int)(threadIdx.x & 63u) >= off) incl += up;
      }
      if ((threadIdx.x & 63u) == 63u) sWave[threadIdx.x >> 6] = incl;
      __syncthreads();
      unsigned running = incl - sum;
      for (unsigned w = 0; w < (threadIdx.x >> 6); w++) running += sWave[w];
#pragma unroll
      for (unsigned j = 0; j < kPer; j++) {
        const unsigned e = threadIdx.x * kPer + j;
        if (e < bins) {
          sLoc[e] = running;
          sDst[e] = start[j] + got[j];
        }
        running += v[j];
      }
    }
    __syncthreads();
#pragma unroll
    for (unsigned j = 0; j < kWideItems; j++)
      if (where[j] != 0xFFFFFFFFu) sRec[sLoc[where[j] >> 16] + (where[j] & 0xFFFFu)] = rec[j];
    __syncthreads();
    for (unsigned b = (threadIdx.x >> 3); b < bins; b += kPartitionThreads / 8u) { /* 8 lanes per bucket: runs of 16-byte records */
      const unsigned count = sCnt[b], loc = sLoc[b];
      if (count != 0u && count <= kLongRun) {
        ulonglong2 *dst = (ulonglong2 *)(out + sDst[b]);
        for (unsigned j = threadIdx.x & 7u; j < count; j += 8u) dst[j] = sRec[loc + j];
      } else if (count > kLongRun && (threadIdx.x & 7u) == 0u) {
        sLong[atomicAdd(&sNumLong, 1u)] = b; /* a sorted batch: left to the whole workgroup, as in partitionKernel */
      }
    }
    __syncthreads();
    const unsigned numLong = sNumLong;
    for (unsigned k = 0; k < numLong; k++) {
      const unsigned b = sLong[k], count = sCnt[b], loc = sLoc[b];
      ulonglong2 *dst = (ulonglong2 *)(out + sDst[b]);
      for (unsigned j = threadIdx.x; j < count; j += kPartitionThreads) dst[j] = sRec[loc + j];
    }
    __syncthreads();
    if (threadIdx.x == 0) sNumLong = 0u;
  }
}

/* workgroup size: 256 threads for every variant (the pair variant keeps the 32-bit superblock bases of the pair image in
 * dynamic LDS, 64 B per 2^23 positions = 24 KB for a GRCh38-sized index, which limits it to 6 workgroups per CU;
 * 512-thread workgroups with half the copies of that table measured slower) */
constexpr int orderedThreads(bool pair) { return pair ? 256 : kThreads; }

/*
 * Instrumented launch (TOUCH; awfmGpuSearchHitsLineTally, not for timing): one bit per (search level, 128-B line) the
 * kernel reads -- the batch's COMPULSORY memory traffic: a line a level needs has to come from HBM at least once, however
 * well the order keeps the re-reads in the L2.  Level = characters of the k-mer already consumed when the step is taken.
 */
constexpr unsigned kTouchLevels = 32;
struct OrderTouch {
  unsigned long long *seedLines; /* bit per 128-B line of the index's seed table */
  unsigned long long *deepLines; /* ... of the deeper device-only table */
  unsigned long long *pairLines; /* [level][pairWords]: bit per pair block (one line each) */
  unsigned long long *nucLines;  /* [level][nucWords]: bit per line of the one-letter image (two 64-B blocks) */
  unsigned long long pairWords, nucWords;
  unsigned long long *hits; /* k-mers that stored a result */
};

/* mixed-length batches: the sample's word that says whether the batch went to mixedLookupSearchKernel instead (the other
 * variants carry no such argument: the fixed-length kernels keep their registers) */
template <bool VARLEN>
struct OrderSkip {
  __device__ __forceinline__ bool chosen() const { return false; }
};
template <>
struct OrderSkip<true> {
  const unsigned *sampleAlive = nullptr;
  unsigned samples = 0;
  __device__ __forceinline__ bool chosen() const { return lookupChosen(sampleAlive, samples, false); }
};

constexpr unsigned kTicketGroups = 4; /* ticket counters per XCD and wave slot */
template <int G, bool NARROW, bool VARLEN, bool PAIR = false, bool TOUCH = false, bool BUCKET = false,
          bool LIST = false /* bucketed records + the list of hits: a wave collects its hits (below) */>
/* registers: 8 waves per SIMD (64 VGPRs) for the one-step variants; the mixed-length, the pair and the bucketed variants get
 * 72 (7 waves) -- the bucketed pair variant, which carries the next chunk's codes, query number and table entry as well,
 * spills two registers there (six in its LIST twin, which collects a wave's hits for the list) and is still the faster build since most k-mers of a batch without hits end at the deeper
 * table (10^8 random 21-mers 2.31-2.38 against 2.57-2.59 ms with 80 registers and 6 waves; planted 5.17 against 5.25) --;
 * the 64-bit pair variants, the instrumented variant and the wide two-lane measurement variant get 80 (6 waves). */
__global__ void __launch_bounds__(orderedThreads(PAIR)) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(G >= 2 ? ((PAIR && !NARROW) || TOUCH || (G == 2 && !NARROW) ? 6 : (VARLEN || PAIR || BUCKET ? 7 : 8)) : 2, 8)))
    orderedSearchKernel(const DevIndex ix, const void *__restrict__ recs, const unsigned long long numRecs,
                        const unsigned *__restrict__ generalCount, const unsigned len, const unsigned depth,
                        const ulonglong2 *__restrict__ table, ulonglong2 *__restrict__ ranges,
                        unsigned *__restrict__ counts, unsigned *__restrict__ tickets,
                        const OrderTouch touch = OrderTouch(), const unsigned *__restrict__ bucketStart = nullptr,
                        const BucketFormat bucketFmt = BucketFormat(), const SparseOut sparse = SparseOut(),
                        const unsigned chunksPerTicket = 1u, const OrderSkip<VARLEN> skip = OrderSkip<VARLEN>()) {
  static_assert(!BUCKET || !VARLEN, "bucketed records are the 8-byte records of fixed-length batches");
  /* a mixed-length batch that the sample gave to mixedLookupSearchKernel: no records were written (uniform) */
  if (skip.chosen()) return;
  constexpr bool AHEAD = BUCKET; /* the next chunk's table entry is requested a chunk ahead */
  constexpr int S = (int)kSlices / G;
  typedef typename PositionType<NARROW>::type pos_t;
  __shared__ unsigned long long sC[24];
  __shared__ unsigned sMask[(kBlockMask + 1) * kSlices];
  __shared__ unsigned long long sSuper[!NARROW ? kMaxNucSuper * 4 : 1];
  /* PAIR (G = 4): two steps per block read through the pair image (awfm_pair.h); its superblock bases are dynamic LDS */
  __shared__ unsigned long long sPairC[PAIR ? 16 : 1];
  extern __shared__ unsigned sPairSuper[];
  static_assert(!PAIR || G == 4, "pair steps are written for 4 lanes per query");
  /* the list of hits (SparseOut::count): a wave collects its hits here and appends them kHitBuffer at a time, or when it is
   * done.  One returning atomic per wave round with a hit is 7 * 10^4 atomics on one word for 10^8 random 21-mers, and a
   * word takes 88 per microsecond: 0.8 ms -- unnoticed inside a 2.3 ms kernel, twice the kernel once encodeLookupKernel had
   * left it 4 * 10^6 k-mers (0.78 against 0.35 ms with dense results) */
  static_assert(!LIST || BUCKET, "the collected list belongs to the bucketed variant (the one that runs after encodeLookupKernel)");
  constexpr unsigned kHitBuffer = LIST ? 32 : 1;
  __shared__ unsigned sHitKmers[LIST ? orderedThreads(PAIR) / 64 : 1][kHitBuffer];
  __shared__ unsigned long long sHitRanges[LIST ? orderedThreads(PAIR) / 64 : 1][kHitBuffer][2];
  unsigned hitFill = 0; /* wave-uniform */
  /* the records this launch covers: the whole order, or the buckets [firstBucket, endBucket) of it */
  const unsigned long long coveredFirst = BUCKET && bucketFmt.endBucket != 0u ? (unsigned long long)bucketStart[bucketFmt.firstBucket] : 0ull;
  const unsigned long long coveredEnd = BUCKET ? (unsigned long long)bucketStart[bucketFmt.endBucket != 0u ? bucketFmt.endBucket : 1u << bucketFmt.bucketBits]
                                               : numRecs - (unsigned long long)*generalCount;
  /* nothing to search (the batch ended in lookupSearchKernel): not even the tables are staged */
  if (BUCKET && coveredEnd == coveredFirst) return;
  /* ... and what the waves of a workgroup still hold when they are done goes out in ONE reservation, made by the wave that
   * finishes last: the waves of the grid end together, and 7168 of them each taking a returning atomic on the list's counter
   * were a tail of 60-80 us on a kernel that searched 5 * 10^5 k-mers (a word takes 88 atomics per microsecond) */
  __shared__ unsigned sHitLeft[LIST ? orderedThreads(PAIR) / 64 : 1], sWavesDone;
  if (LIST && threadIdx.x == 0) sWavesDone = 0u;
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  stageMaskTable(sMask);
  nucStageSuper<NARROW>(ix, sSuper);
  if (PAIR) pairStageTables<NARROW, 16u>(ix, sPairC, sPairSuper);
  __syncthreads();

  /* A record as loaded: {codes, query number | length << 32} (16 bytes), or BUCKET the 8-byte record.  Every loaded value
   * lands in the register it is later read from (no widening, no copy: either would be waited for right after the load)
   * and is taken apart one iteration later. */
  struct Raw {
    unsigned long long a, b; /* 16-byte records: the two words; BUCKET: a = record */
  };
  auto readRecord = [&](unsigned long long at, Raw &r) {
    if (BUCKET) {
      r.a = ((const unsigned long long *)recs)[at];
    } else {
      const ulonglong2 w = *(const ulonglong2 *)((const QueryRec *)recs + at);
      r.a = w.x;
      r.b = w.y;
    }
  };
  const unsigned gl = threadIdx.x % G;
  const unsigned firstSlice = gl * S;
  auto markLine = [&](unsigned long long *bits, unsigned long long line) {
    if (TOUCH && gl == 0) atomicOr(bits + (line >> 6), 1ull << (line & 63ull));
  };
  /* the lines a one-letter step / a pair step from the range {sp, ep} reads */
  auto touchNuc = [&](unsigned level, unsigned long long sp, unsigned long long ep) {
    if (TOUCH) {
      unsigned long long *bits = touch.nucLines + (level < kTouchLevels ? level : kTouchLevels - 1u) * touch.nucWords;
      markLine(bits, ((sp - 1ull) >> kBlockShift) >> 1);
      markLine(bits, (ep >> kBlockShift) >> 1);
    }
  };
  auto touchPair = [&](unsigned level, unsigned long long sp, unsigned long long ep) {
    if (TOUCH) {
      unsigned long long *bits = touch.pairLines + (level < kTouchLevels ? level : kTouchLevels - 1u) * touch.pairWords;
      markLine(bits, (sp - 1ull) >> kBlockShift);
      markLine(bits, ep >> kBlockShift);
    }
  };
  /* the records the fast path covers come first in the order; each XCD takes a contiguous eighth of them */
  /* (bucketed records: what lies before the last bin -- the batch without the k-mers of the general kernel, or the k-mers
   * encodeLookupKernel kept) */
  const unsigned long long covered = coveredEnd - coveredFirst;
  const unsigned xcds = (gridDim.x & 7u) == 0u ? 8u : 1u; /* (workgroup b runs on XCD b % 8 under round-robin dispatch) */
  const unsigned xcd = xcds == 8u ? (blockIdx.x & 7u) : 0u;
  const unsigned blockInXcd = xcds == 8u ? (blockIdx.x >> 3) : blockIdx.x;
  const unsigned long long share = (covered + xcds - 1ull) / xcds;
  const unsigned long long begin = coveredFirst + share * xcd;
  const unsigned long long end = begin + share < coveredEnd ? begin + share : coveredEnd;
  /* The waves of an XCD take chunks of 64/G consecutive records from ticket counters instead of a fixed stride:
   * free-running waves drift apart, and with a fixed stride the records in flight on an XCD would then span
   * many more buckets than its L2 holds the blocks of (8.7 ms with the stride, 5.8 ms with tickets).  Wave w of
   * a workgroup draws from counter w of its XCD and takes chunk 4*ticket + w, so a counter sees a quarter of the
   * traffic and no barrier ties the waves of a workgroup together.  A ticket is drawn two chunks ahead and read
   * at the end of an iteration, so its latency and the record read hide behind the current chunk.  (Requesting the
   * table entry a chunk ahead as well -- no dependent round left for the seed lookup -- changed nothing: 4.37-4.49 ms
   * either way per 10^8 random 21-mers, and the extra registers cost the one-letter variant a wave per SIMD.  The
   * kernel's time follows the number of block reads that miss the L2, not the length of a wave's chain.  Nor does it
   * follow the number of rounds a wave waits through: a variant that refilled every group of 4 lanes with the next
   * record as soon as its k-mer was done -- record and table entry staged two k-mers ahead, one memory round per
   * iteration for all 16 groups, 2.55 rounds per 16 k-mers instead of 1 + 3.5 -- gave the same ranges in 5.20-5.24 ms
   * against 4.41-4.45 ms on the same box, 9.2 against 9.1-9.4 ms planted.) */
  constexpr unsigned kWaves = orderedThreads(PAIR) / 64, kChunk = 64 / G;
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  auto flushHits = [&]() { /* wave-uniform: the wave's collected hits go to the list, one reservation for all of them */
    if (hitFill != 0u) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      unsigned listBase = 0;
      if (lane == 0) listBase = atomicAdd(sparse.count, hitFill);
      listBase = (unsigned)__builtin_amdgcn_readfirstlane((int)listBase);
      if (lane < hitFill && listBase + lane < sparse.cap) {
        const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        sparse.kmers[listBase + lane] = sHitKmers[w][lane];
        sparse.ranges[listBase + lane] = make_ulonglong2(sHitRanges[w][lane][0], sHitRanges[w][lane][1]);
      }
      __builtin_amdgcn_wave_barrier();
      hitFill = 0;
    }
  };
  /* kTicketGroups x kWaves counters per XCD: wave w of workgroup b draws from counter (b % kTicketGroups, w) */
  constexpr unsigned kCounters = kTicketGroups * kWaves;
  const unsigned counter = (blockInXcd % kTicketGroups) * kWaves + wave;
  unsigned *ticket = tickets + (xcd * kCounters + counter) * 64u; /* 256 bytes apart */
  /* A ticket is worth chunksPerTicket consecutive chunks of the wave (one returning atomic on a counter that 48-64 waves
   * share, per ticket): the next ticket is drawn when the wave starts on the current one and read when that one is used
   * up, so the atomic has chunksPerTicket iterations to come back.  (One chunk per ticket: 52 atomics per microsecond and
   * counter, the order of what one word takes (MI355X_MICROARCH.md: 88); an iteration could not end before its own
   * atomic had come back, and that round trip, not the k-mers' memory rounds, set the kernel's time: a build that cut
   * the search short after one or two pair steps was no faster.) */
  auto ticketBase = [&](unsigned t) -> unsigned long long {
    return begin + ((unsigned long long)t * kCounters + counter) * chunksPerTicket * kChunk;
  };
  unsigned drawn = 0;
  if (lane == 0) drawn = atomicAdd(ticket, 1u);
  unsigned long long ticketAt = ticketBase((unsigned)__builtin_amdgcn_readfirstlane((int)drawn));
  unsigned sub = 0;
  if (lane == 0) drawn = atomicAdd(ticket, 1u); /* the ticket after */
  auto nextChunk = [&]() -> unsigned long long { /* wave-uniform */
    if (sub == chunksPerTicket) {
      ticketAt = ticketBase((unsigned)__builtin_amdgcn_readfirstlane((int)drawn));
      sub = 0;
      if (lane == 0) drawn = atomicAdd(ticket, 1u);
    }
    return ticketAt + (unsigned long long)(sub++) * kChunk;
  };
  unsigned long long base = nextChunk();
  unsigned long long baseNext = nextChunk();

  const unsigned long long tableMask = (1ull << (2u * depth)) - 1ull;
  Raw raw = {0ull, 0ull}; /* the prefetched record */
  if (base + lane / G < end) readRecord(base + lane / G, raw);
  /* BUCKET: a record does not hold the bits its bucket stands for; the bucket of a position is where bucketStart says.
   * The wave keeps the bucket of the chunk it looked at last (its chunks come in increasing order): wave-uniform, scalar
   * loads. */
  /* (the buckets this launch covers end at endBucket: behind it an array that holds a share of the order only has every start
   * equal to its length, and a wave looking for the bucket of the share's last chunk walked through all of them -- 1792 dependent
   * scalar loads, 0.18 ms behind a 0.67-ms search of an eighth of the order) */
  const unsigned numBuckets = BUCKET ? (bucketFmt.endBucket != 0u ? bucketFmt.endBucket : 1u << bucketFmt.bucketBits) : 0u;
  unsigned waveBucket = 0, waveNext = 0;
  if (BUCKET && base < end) {
    unsigned lo = 0, hi = numBuckets - 1u; /* the last bucket that starts at or before `base` */
    while (lo < hi) {
      const unsigned mid = (lo + hi + 1u) >> 1;
      if ((unsigned long long)bucketStart[mid] <= base) lo = mid;
      else hi = mid - 1u;
    }
    waveBucket = lo;
    waveNext = bucketStart[lo + 1u];
  }
  /* bucket of this lane's record in the chunk that starts at chunkAt (not before the chunk asked for last) */
  auto laneBucket = [&](unsigned long long chunkAt) -> unsigned {
    while ((unsigned long long)waveNext <= chunkAt && waveBucket + 1u < numBuckets) {
      waveBucket++;
      waveNext = bucketStart[waveBucket + 1u];
    }
    unsigned mine = waveBucket, nb = waveBucket, ns = waveNext; /* the chunk may reach into the following bucket(s) */
    while ((unsigned long long)ns <= chunkAt + (kChunk - 1u) && nb + 1u < numBuckets) {
      nb++;
      mine = chunkAt + lane / G >= (unsigned long long)ns ? nb : mine;
      ns = bucketStart[nb + 1u];
    }
    return mine;
  };
  /* With the next-step bits of the deeper table (DevIndex::deepNext) a k-mer whose first pair step would empty the range
   * ends at the table, without the block read (fixed-length batches with a pair step to come; only hits are reported):
   * 10^8 random 21-mers, depth 16: 8.1 instead of 25 million distinct pair-block lines, 2.55-2.70 against 2.9-3.0 ms.
   * (Taking the entry apart where it is used instead of where it is loaded: 2.86 ms.  With the bits the kernel reads
   * 13.5 GB in 2.34 ms = 5.8 TB/s of random 128-byte lines, 10.7 GB of them table lines: a variant that gave every LANE a
   * record and a table entry -- 64 lookups per wave instruction -- and handed the k-mers still alive to the groups of 4
   * lanes, 16 per round (ds_permute / ds_bpermute), measured 2.44 ms, and 5.99 against 5.22 ms on planted k-mers.) */
  const bool rawEntries = table == ix.deepSeed && ix.deepNarrow != 0u; /* uniform: 8-byte entries, whose format deepSeedOpen reads */
  const bool dropByNext = PAIR && !VARLEN && rawEntries && ix.deepNext != 0u && len >= depth + 2u;
  auto tableEntry = [&](unsigned long long codes) -> ulonglong2 {
    const unsigned long long at = codes & tableMask;
    if (rawEntries) {
      unsigned next16;
      ulonglong2 r = deepSeedOpen(ix, at, ((const uint2 *)ix.deepSeed)[at], &next16);
      if (dropByNext && ((next16 >> ((unsigned)(codes >> (2u * depth)) & 15u)) & 1u) == 0u) r = make_ulonglong2(1ull, 0ull);
      return r;
    }
    return table == ix.deepSeed ? deepSeedEntry(ix, at) : table[at];
  };
  auto touchTable = [&](unsigned long long at) {
    if (TOUCH) markLine(table == ix.seed ? touch.seedLines : touch.deepLines, at >> (table == ix.deepSeed && ix.deepNarrow ? 4 : 3));
  };
  /* BUCKET runs one chunk further ahead than the other formats: at the top of an iteration the record of the NEXT chunk
   * is taken apart and its table entry requested, so that an entry has a whole iteration to arrive and its wait merges
   * with the first block reads of the chunk before it.  (With the deeper table most entries are compulsory misses: one
   * of the 3.5 memory rounds of a chunk of random 21-mers.  10^8 random 21-mers: 3.77-3.80 against 3.87-3.93 ms.)
   * `raw` then holds the record of the chunk after `base`. */
  unsigned long long codesCur = 0, baseNext2 = 0;
  unsigned indexCur = 0;
  pos_t entrySpCur = 1, entryEpCur = 0; /* the next chunk's table entry, in position width */
  if (AHEAD) {
    baseNext2 = nextChunk();
    if (base < end) {
      const unsigned mine = laneBucket(base);
      codesCur = bucketCodes(bucketFmt, mine, raw.a >> bucketFmt.indexBits);
      indexCur = (unsigned)(raw.a & ((1ull << bucketFmt.indexBits) - 1ull));
      if (base + lane / G < end) {
        const ulonglong2 r = tableEntry(codesCur);
        entrySpCur = (pos_t)r.x;
        entryEpCur = (pos_t)r.y;
        touchTable(codesCur & tableMask);
      }
    }
    if (baseNext + lane / G < end) readRecord(baseNext + lane / G, raw);
  }
  while (base < end) { /* wave-uniform */
    const unsigned long long q = base + lane / G;
    const bool live = q < end;
    pos_t sp = 1, ep = 0;
    int pos = -1;
    unsigned long long rem;
    unsigned long long codes;
    unsigned index;
    pos_t entrySp = 1, entryEp = 0;
    if (AHEAD) {
      /* this chunk: taken apart an iteration ago, its entry requested then */
      codes = codesCur;
      index = indexCur;
      entrySp = entrySpCur;
      entryEp = entryEpCur;
      asm volatile("" : "+v"(entrySp), "+v"(entryEp));
      /* the next chunk: its record (requested an iteration ago) is taken apart, its table entry and the record of the
       * chunk after it are requested */
      entrySpCur = 1;
      entryEpCur = 0;
      if (baseNext < end) { /* wave-uniform */
        const unsigned mine = laneBucket(baseNext);
        asm volatile("" : "+v"(raw.a)::"memory");
        codesCur = bucketCodes(bucketFmt, mine, raw.a >> bucketFmt.indexBits);
        indexCur = (unsigned)(raw.a & ((1ull << bucketFmt.indexBits) - 1ull));
        if (baseNext + lane / G < end) {
          const ulonglong2 r = tableEntry(codesCur);
          entrySpCur = (pos_t)r.x;
          entryEpCur = (pos_t)r.y;
          touchTable(codesCur & tableMask);
        }
      }
      if (baseNext2 + lane / G < end) readRecord(baseNext2 + lane / G, raw);
    } else {
    /* the record fetched an iteration ago is taken apart BEFORE anything new is issued: a wait placed after the
     * atomic below would also wait for that atomic */
    asm volatile("" : "+v"(raw.a), "+v"(raw.b)::"memory");
    codes = BUCKET ? bucketCodes(bucketFmt, laneBucket(base), raw.a >> bucketFmt.indexBits) : raw.a;
    index = BUCKET ? (unsigned)(raw.a & ((1ull << bucketFmt.indexBits) - 1ull)) : (unsigned)raw.b;
    }
    const unsigned myLen = VARLEN ? (unsigned)(raw.b >> 32) : len; /* before `raw` is overwritten by the prefetch */
    /* ---- seed (ref src/AwFmKmerTable.c:4-51): the index table, or the deeper device-only one ---- */
    if (!AHEAD) {
    /* fixed length: the table entry is requested FIRST, so that the wait for it (loads return in order) is not also a
     * wait for the ticket atomic and the record prefetch issued below */
    if (!VARLEN && live) {
      const ulonglong2 r = tableEntry(codes);
      entrySp = (pos_t)r.x;
      entryEp = (pos_t)r.y;
      touchTable(codes & tableMask);
    }
    /* fetch the next chunk's record */
    if (baseNext + lane / G < end) readRecord(baseNext + lane / G, raw);
    }
    if (!VARLEN) {
      if (live) {
        sp = entrySp;
        ep = entryEp;
        pos = (int)(len - depth) - 1;
      }
      rem = codes >> (2u * depth); /* code of character `pos` in bits 1..0 */
    } else {
      /* per-query length (in the record) and start: deeper table, seed table, or the letter range of the last
       * character (ref src/AwFmSearch.c:485-502) */
      const unsigned myDepth = orderStartDepth(myLen, ix.seedK, ix.deepK);
      if (live) {
        if (myDepth != 0u) {
          const unsigned long long at = codes & ((1ull << (2u * myDepth)) - 1ull);
          ulonglong2 r;
          if (myDepth == ix.seedK) {
            r = ix.seed[at];
          } else if (ix.deepNarrow != 0u) {
            unsigned next16;
            r = deepSeedOpen(ix, at, ((const uint2 *)ix.deepSeed)[at], &next16);
            /* the first step from the deeper table is a pair step (below): a clear bit ends the k-mer here */
            if (PAIR && ix.deepNext != 0u && myLen >= myDepth + 2u && ((next16 >> ((unsigned)(codes >> (2u * myDepth)) & 15u)) & 1u) == 0u)
              r = make_ulonglong2(1ull, 0ull);
          } else {
            r = deepSeedEntry(ix, at);
          }
          if (TOUCH) markLine(myDepth == ix.seedK ? touch.seedLines : touch.deepLines, at >> (myDepth != ix.seedK && ix.deepNarrow ? 4 : 3));
          sp = (pos_t)r.x;
          ep = (pos_t)r.y;
          pos = (int)(myLen - myDepth) - 1;
        } else {
          const unsigned a = (unsigned)codes & 3u;
          sp = (pos_t)sC[a];
          ep = (pos_t)(sC[a + 1] - 1ull);
          pos = (int)myLen - 2;
        }
      }
      rem = codes >> (myDepth != 0u ? 2u * myDepth : 2u);
    }

    /* ---- extension (ref src/AwFmParallelSearch.c:273-313) ---- */
    const int lastPos = (int)myLen - 1; /* TOUCH: level of a step = lastPos - pos = characters consumed before it */
    if (PAIR) {
      /* two characters per block read.  Only hits are reported, so it does not matter at which of the two steps a
       * range without hits became empty.  An odd step is taken alone and last: few k-mers are still alive then, and the
       * first step from the deeper table is the pair step its next-step bits speak about. */
      while (pos >= 1 && sp <= ep) {
        const unsigned c2 = (unsigned)rem & 3u, c1 = (unsigned)(rem >> 2) & 3u;
        touchPair((unsigned)(lastPos - pos), sp, ep);
        if (pairSearchStep<NARROW>(ix, sPairC, sPairSuper, sMask, gl, c1 * 4u + c2, sp, ep) == kPairFlagged) {
          /* a block with an ambiguity letter or the sentinel: letter by letter through the one-letter image */
          touchNuc((unsigned)(lastPos - pos), sp, ep);
          nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, c2, sp, ep);
          if (sp <= ep) {
            touchNuc((unsigned)(lastPos - pos) + 1u, sp, ep);
            nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, c1, sp, ep);
          }
        }
        pos -= 2;
        rem >>= 4;
      }
      if (pos == 0 && sp <= ep) {
        touchNuc((unsigned)(lastPos - pos), sp, ep);
        nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, (unsigned)rem & 3u, sp, ep);
        pos--;
      }
    } else {
      while (pos >= 0 && sp <= ep) {
        touchNuc((unsigned)(lastPos - pos), sp, ep);
        nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, (unsigned)rem & 3u, sp, ep);
        pos--;
        rem >>= 2;
      }
    }
    if (TOUCH && base + lane / G < end && gl == 0 && sp <= ep) atomicAdd(touch.hits, 1ull);
    { /* (position in the order and liveness are recomputed here rather than kept in registers across the steps) */
      const unsigned long long at = base + lane / G;
      const bool mine = at < end && gl == 0;
      if (sparse.count) { /* kernel argument: uniform */
        const bool hit = mine && sp <= ep;
        if (!LIST) sparseAppend(sparse, hit, index, (unsigned long long)sp, (unsigned long long)ep);
        const unsigned long long hitMask = LIST ? __ballot(hit) : 0ull;
        if (hitMask != 0ull) { /* wave-uniform; at most 64 / G hits a round */
          const unsigned hits = (unsigned)__builtin_amdgcn_readfirstlane((int)__popcll(hitMask));
          if (hit) { /* there is room: the buffer is emptied below whenever a round's worth might not fit */
            const unsigned at = hitFill + (unsigned)__popcll(hitMask & ((1ull << (threadIdx.x & 63u)) - 1ull));
            const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            sHitKmers[w][at] = index;
            sHitRanges[w][at][0] = (unsigned long long)sp;
            sHitRanges[w][at][1] = (unsigned long long)ep;
          }
          hitFill = (unsigned)__builtin_amdgcn_readfirstlane((int)(hitFill + hits));
        }
      } else if (sparse.kmers) { /* results in search order: entry `at`, whatever the outcome */
        if (mine && sparse.ranges) {
          sparse.kmers[at - coveredFirst] = index; /* (a share of the order: its entries from 0) */
          sparse.ranges[at - coveredFirst] = sp <= ep ? make_ulonglong2((unsigned long long)sp, (unsigned long long)ep) : make_ulonglong2(1ull, 0ull);
          /* the counts in the same order when they are asked for (awfmGpuSearchHitsInOrderCounts): the scan that follows reads
           * 4 instead of 16 bytes per k-mer */
          if (counts) counts[at - coveredFirst] = sp <= ep ? (unsigned)(ep - sp + (pos_t)1) : 0u;
        } else if (mine) { /* counts only (awfm_count_order_kernel.h): {k-mer number, count}, taken home by two passes */
          ((uint2 *)sparse.kmers)[at - coveredFirst] = make_uint2(index, sp <= ep ? (unsigned)(ep - sp + (pos_t)1) : 0u);
        }
      } else if (mine && sp <= ep) {
        if (ranges) ranges[index] = make_ulonglong2((unsigned long long)sp, (unsigned long long)ep);
        if (counts) counts[index] = (unsigned)(ep - sp + (pos_t)1);
      }
    }
    if (LIST && hitFill + 64u / G > kHitBuffer) flushHits();
    base = baseNext;
    if (AHEAD) {
      baseNext = baseNext2;
      baseNext2 = nextChunk();
    } else {
      baseNext = nextChunk();
    }
  }
  if (LIST) {
    const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (lane == 0) sHitLeft[w] = hitFill;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    unsigned arrived = 0;
    if (lane == 0) arrived = atomicAdd(&sWavesDone, 1u);
    arrived = (unsigned)__builtin_amdgcn_readfirstlane((int)arrived);
    if (arrived == kWaves - 1u) { /* wave-uniform: every wave's leftovers are in LDS */
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      unsigned total = 0;
      for (unsigned v = 0; v < kWaves; v++) total += sHitLeft[v];
      if (total != 0u) {
        unsigned listBase = 0;
        if (lane == 0) listBase = atomicAdd(sparse.count, total);
        listBase = (unsigned)__builtin_amdgcn_readfirstlane((int)listBase);
        unsigned before = 0;
        for (unsigned v = 0; v < kWaves; v++) {
          const unsigned n = sHitLeft[v]; /* at most kHitBuffer <= 64 */
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


/* ---- the "next pair step" bits of the deeper table (DevIndex::deepNext) ----
 * One group of 4 lanes per entry {sp, length} with length > 0: the 16 pair steps the search kernel could take from that
 * range -- the same device functions, flagged blocks through the one-letter image as there -- and bit c of next16 set
 * when the range after step c still holds a position.  A length that does not fit 16 bits goes to bigBySp[sp >> 15]
 * (awfm_device.h: deepBigLength).  Persistent grid, chunks of 16 entries per wave at a fixed stride. */
/* NARROW: the image runs 32-bit positions and the entries are {sp, length} (format 1), rewritten as {sp, length16 | next16 << 16};
 * otherwise (round 6) they are format 2 -- sp36 | length12 | next16, the long lengths in `big` already, DevIndex::deepBigBySp of
 * `ix` set by the caller -- and only the sixteen bits are rewritten. */
template <bool NARROW>
__global__ void __launch_bounds__(orderedThreads(true)) __attribute__((amdgpu_num_sgpr(80)))
    deepNextKernel(const DevIndex ix, uint2 *__restrict__ table, const unsigned long long numEntries,
                   unsigned *__restrict__ bigBySp, unsigned *__restrict__ numBig) {
  constexpr int G = 4;
  constexpr int S = (int)kSlices / G;
  typedef typename PositionType<NARROW>::type pos_t;
  __shared__ unsigned long long sC[24];
  __shared__ unsigned sMask[(kBlockMask + 1) * kSlices];
  __shared__ unsigned long long sSuper[NARROW ? 1 : kMaxNucSuper * 4];
  __shared__ unsigned long long sPairC[16];
  extern __shared__ unsigned sPairSuper[];
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  stageMaskTable(sMask);
  nucStageSuper<NARROW>(ix, sSuper);
  pairStageTables<NARROW, 16u>(ix, sPairC, sPairSuper);
  __syncthreads();
  const unsigned gl = threadIdx.x % G, lane = threadIdx.x & 63u;
  const unsigned firstSlice = gl * S;
  constexpr unsigned long long kChunk = 64 / G;
  /* waves take chunks of 16 entries at a fixed stride (a shared cursor -- 2.7 * 10^8 atomics on one word for a depth-16
   * table -- took 3.2 s at the 88 atomics per microsecond a word serves) */
  const unsigned long long waveStride = (unsigned long long)gridDim.x * (blockDim.x / 64u) * kChunk;
  for (unsigned long long base = ((unsigned long long)blockIdx.x * (blockDim.x / 64u) + threadIdx.x / 64u) * kChunk; base < numEntries;
       base += waveStride) {
    const unsigned long long at = base + lane / G;
    const uint2 e = at < numEntries ? table[at] : make_uint2(1u, 0u);
    const bool some = NARROW ? e.y != 0u : (e.y & (kDeepWideLengthMask << 4)) != 0u;
    if (some) { /* (empty: no such deepK-mer, and its entry is final as it is) whole groups of 4 lanes */
      ulonglong2 r;
      if (NARROW) r = make_ulonglong2((unsigned long long)e.x, (unsigned long long)e.x + e.y - 1ull);
      else r = deepSeedOpen(ix, at, e, nullptr);
      unsigned next16 = 0;
      for (unsigned code = 0; code < 16u; code++) {
        pos_t sp = (pos_t)r.x, ep = (pos_t)r.y;
        if (pairSearchStep<NARROW>(ix, sPairC, sPairSuper, sMask, gl, code, sp, ep) == kPairFlagged) {
          nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, code & 3u, sp, ep);
          if (sp <= ep) nucFastStep<G, NARROW>(ix, sC, sSuper, sMask, firstSlice, code >> 2, sp, ep);
        }
        if (sp <= ep) next16 |= 1u << code;
      }
      if (gl == 0) {
        if (NARROW) {
          if (e.y >= 0xFFFFu) { /* (awfm_device.h: deepBigLength) */
            bigBySp[e.x >> kDeepBigShift] = e.y;
            atomicAdd(numBig, 1u);
          }
          table[at] = make_uint2(e.x, (e.y < 0xFFFFu ? e.y : 0xFFFFu) | next16 << 16);
        } else {
          if (((e.y >> 4) & kDeepWideLengthMask) == kDeepWideLengthMask) atomicAdd(numBig, 1u);
          table[at] = make_uint2(e.x, (e.y & 0xFFFFu) | next16 << 16);
        }
      }
    }
  }
}

/* The amino twin (awfm_device.h: kAminoDeepLengthBits): every non-empty entry {sp, length} of a finished depth-deepK table
 * becomes {sp, length12 | next20 << 12}, bit c set when the range still holds a position after one more step with letter c.
 * One group of 4 lanes per entry, 20 steps each (aminoStepAny: the search kernels' step); lengths of 4095 and more go to
 * bigBySp[sp >> 11]. */
template <bool NARROW>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_num_sgpr(80)))
    aminoDeepNextKernel(const DevIndex ix, uint2 *__restrict__ table, const unsigned long long numEntries, unsigned *__restrict__ bigBySp,
                        unsigned *__restrict__ numBig) {
  constexpr int G = 4;
  __shared__ unsigned long long sC[24];
  __shared__ AminoShared sAmino;
  __shared__ unsigned sMask[(kBlockMask + 1) * kSlices];
  if (threadIdx.x < 24) sC[threadIdx.x] = ix.prefixSums[threadIdx.x];
  aminoStageTables(sAmino);
  stageMaskTable(sMask);
  __syncthreads();
  const unsigned gl = threadIdx.x % G, lane = threadIdx.x & 63u;
  constexpr unsigned long long kChunk = 64 / G;
  const unsigned long long waveStride = (unsigned long long)gridDim.x * (blockDim.x / 64u) * kChunk;
  for (unsigned long long base = ((unsigned long long)blockIdx.x * (blockDim.x / 64u) + threadIdx.x / 64u) * kChunk; base < numEntries;
       base += waveStride) {
    const unsigned long long at = base + lane / G;
    const uint2 e = at < numEntries ? table[at] : make_uint2(1u, 0u);
    /* NARROW: the finished table is {sp, length}; otherwise aminoWidePack's entries with every bit set and the long lengths
     * already beside them (ix.deepNarrow == 2, ix.deepBigBySp: aminoDeepSeedLevelKernel<2> wrote both) */
    const unsigned long long first = NARROW ? (unsigned long long)e.x : deepWideSp(e);
    const unsigned long long length = NARROW ? (unsigned long long)e.y : aminoDeepLength(ix, e);
    if (length != 0ull) { /* whole groups of 4 lanes */
      unsigned next20 = 0;
      for (unsigned letter = 0; letter < 20u; letter++) {
        typename PositionType<NARROW>::type sp = (typename PositionType<NARROW>::type)first, ep = (typename PositionType<NARROW>::type)(first + length - 1ull);
        aminoStepAny<G, NARROW>(ix, sC, sAmino, sMask, gl, letter, sp, ep);
        if (sp <= ep) next20 |= 1u << letter;
      }
      if (gl == 0) {
        if (NARROW) {
          if (e.y >= kAminoDeepLengthMask) {
            bigBySp[e.x >> kAminoDeepBigShift] = e.y;
            atomicAdd(numBig, 1u);
          }
          table[at] = make_uint2(e.x, (e.y < kAminoDeepLengthMask ? e.y : kAminoDeepLengthMask) | next20 << kAminoDeepLengthBits);
        } else {
          if (length >= kAminoWideLengthMask) atomicAdd(numBig, 1u);
          table[at] = make_uint2(e.x, (e.y & kAminoDeepLengthMask) | next20 << kAminoDeepLengthBits);
        }
      }
    }
  }
}

}  // namespace

#endif
