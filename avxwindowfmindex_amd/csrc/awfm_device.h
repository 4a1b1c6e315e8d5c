/*
 * awfm_device.h -- device-side building blocks shared by awfm_gpu.hip (search /
 * locate) and awfm_gpu_build.hip (index construction): the kernel-argument view
 * of the device image, the rank/step primitives on the device block layouts and the
 * layout conversion kernels.  C++/HIP only; the C ABI is include/awfm_gpu.h.
 *
 * Device block layouts (the host AwFmIndex keeps the reference's 160-B / 352-B blocks of 256 positions,
 * ref src/AwFmIndex.h:55-65; the device image is re-laid-out once per index):
 *
 *   A device block covers 128 BWT positions as 4 slices of 32 positions, for both alphabets.
 *
 *   nucleotide block = 64 B.  Slice k is one 16-B piece {b0, b1, b2, count}: the three bit-plane words of positions
 *     32k..32k+31 and the 32-bit count of letter k (a, c, g, t) before the block, relative to the block's
 *     superblock of 2^32 positions.  A rank reads one 64-B granule, a lane of a 4-lane group holds one piece per
 *     block, and an index of fewer than 2^32 positions has a single superblock with base 0.  The X count is
 *     derived: positions before the block minus A+C+G+T minus the sentinel; '$' is never ranked.
 *
 *   amino block = 128 B = one line.  Slice k is two 16-B pieces {b0,b1,b2,b3} {b4, c01, c23, c45}: the five plane
 *     words and six 16-bit counts (letters 6k..6k+5; 21 letters including z) relative to the block's superblock
 *     of 2^16 positions (512 blocks).
 *
 *   super: absolute 64-bit base counts at superblock starts -- nucleotide super[4*sb + a] (at most 64 superblocks,
 *     copied to LDS by the kernels of images of 2^32 or more positions), amino super[24*sb + a] (24 words per 2^16
 *     positions: 0.6 MB for a Swiss-Prot-sized index, read beside the block).
 *
 * Occ is a function of the BWT only, so the block granularity changes no result, only which bytes a rank reads.
 */
#ifndef AWFM_DEVICE_H
#define AWFM_DEVICE_H

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include <string>

#include "awfm_internal.h"

void awfmGpuSetError(const char *what);
void awfmGpuSetHipError(const char *what, hipError_t e);

namespace {

inline void setError(const char *what, hipError_t e) { awfmGpuSetHipError(what, e); }
inline void setError(const char *what) { awfmGpuSetError(what); }

#define AWFM_HIP_TRY(call, failRc)                      \
  do {                                                  \
    hipError_t err__ = (call);                          \
    if (err__ != hipSuccess) {                          \
      setError(#call, err__);                           \
      return (failRc);                                  \
    }                                                   \
  } while (0)

constexpr unsigned kBlockShift = 7;       /* 128 positions per device block */
constexpr unsigned kBlockMask = 127;
constexpr unsigned kSlices = 4;           /* 32-position slices per device block */
constexpr unsigned kNucSuperShift = 32;   /* positions per nucleotide superblock: 2^32 ($AWFM_GPU_DIAG nuc_super_shift: tests) */
constexpr unsigned kMaxNucSuper = 64;     /* nucleotide images of up to 2^38 positions */
constexpr unsigned kAminoSuperShift = 16; /* positions per amino superblock: 2^16 */
constexpr unsigned kAminoSuperStride = 24;
constexpr unsigned kPairSuperShift = 24;  /* positions per superblock of the pair image: 2^24 (a count BEFORE a block is at most 2^24 - 128: 24 bits) */
constexpr unsigned kPairCountMask = 0xFFFFFFu;
/* where the two 16-B pieces of slice k of pair block blk are (in uint4 units): the four plane pieces fill the first
 * 64-B sector of the line and the four count pieces the second, so each of the two load instructions of a step touches
 * one sector (with plane and count piece of a slice side by side every instruction touched both; measured the same) */
__host__ __device__ inline unsigned long long pairPlanesAt(unsigned long long blk, unsigned k) { return blk * 8ull + k; }
__host__ __device__ inline unsigned long long pairCountsAt(unsigned long long blk, unsigned k) { return blk * 8ull + 4u + k; }
constexpr unsigned kPairSuperStride = 20; /* words per superblock of the pair image: 16 pairs, then the letters a,c,g,t */

/* kernel-argument view of the device image */
struct DevIndex {
  const uint4 *blocks;
  const ulonglong2 *seed;
  const unsigned long long *sa; /* packed sampled SA viewed as 64-bit words */
  unsigned long long bwtLength;
  unsigned long long sentinelPos; /* BWT position holding '$' */
  unsigned long long seedLen;
  const unsigned long long *prefixSums; /* 24 words in device memory */
  const unsigned long long *super;      /* base counts at superblock starts (layouts above) */
  unsigned int numSuper;
  unsigned int nucSuperShift; /* log2 of the positions per nucleotide superblock: 32 (smaller only in tests) */
  unsigned int saRatio;
  unsigned int saShift; /* log2(saRatio) when it is a power of two, else 0xFFFFFFFF */
  unsigned int saWidth;
  unsigned int seedK;
  /* optional device-only deeper seed table (nucleotide): entry of a deepK-mer = the range the reference
   * algorithm reaches after the seed lookup and (deepK - seedK) extension steps that stop at the first
   * invalid range; NULL when not built */
  const ulonglong2 *deepSeed;
  unsigned int deepK;
  /* entries of the deeper table (deepNarrow): 0 -- 16 bytes {sp, ep}; 1 -- 8 bytes {sp, length} on images below 2^32
   * positions -- exact: the stepping stops at the FIRST empty range, and that one is always {sp, sp - 1} (sp = C + Occ(sp' - 1),
   * ep = C + Occ(ep') - 1 from a valid {sp', ep'}), so a length of 0 says all there is to say; sp >= 1 always --; 2 (round 6) --
   * 8 bytes on images of 2^32 .. 2^36 positions, one 64-bit word sp36 | length12 << 36 | next16 << 48 (ref src/AwFmIndex.h:88-91
   * is 64-bit throughout; 288 GB of HBM hold no image beyond 2^36 positions with its tables).  The kernels read the format at
   * run time (uniform), whatever position width they were compiled for. */
  unsigned int deepNarrow;
  /* (format 1) the 8-byte entries are {sp, length16 | next16 << 16} (awfmGpuDeepSeedAddNext, built when the image has pair
   * blocks): bit c of next16 says whether the range is still non-empty after the pair step with code c (the two
   * characters that precede the deepK-mer in a longer k-mer; awfm_pair.h), so a hits-only search drops a k-mer whose
   * bit is clear without reading a block.  A length of 0xFFFF stands for "0xFFFF or more": the exact one is
   * deepBigBySp[sp >> 15] -- the ranges of two entries are disjoint, so two that are 65535 positions long or longer begin
   * 65535 or more apart and never share a window of 2^15 positions: one read instead of round 4's search in a sorted side
   * list (9 dependent reads for the 316 such entries of a genome-shaped text, which 2 % of the k-mers drawn from it hit).
   * Format 2 always has its 16 bits (all set until the pass that computes them has run: deepNext says whether it has), a
   * 12-bit length whose 0xFFF stands for "4095 or more", and 64-bit exact lengths in deepBigBySp viewed as
   * unsigned long long [sp >> 11]. */
  unsigned int deepNext;
  unsigned int numDeepBig; /* how many such entries the table has (reporting) */
  const unsigned int *deepBigBySp;
  /* the tables per k-mer length (awfmGpuBuildLengthTables) of an image whose deeper table has format 2: entries
   * sp36 | length28 << 36, a length of 2^28 - 1 standing for that or more: the exact one is lengthBig[(d - 1) * 512 + (sp >> 27)]
   * for level d (ranges of ONE level are disjoint; those of different levels nest) */
  const unsigned long long *lengthBig;
  /* optional device-only pair image (nucleotide; awfm_pair.h): two backward / LF steps per block read; NULL when
   * not built.  pairSuper32 is the 32-bit copy of the superblock bases the kernels of images below 2^32 positions
   * keep in LDS (kPairSuperStride words per superblock). */
  const uint4 *pairBlocks;
  const unsigned long long *pairSuper;
  const unsigned int *pairSuper32;
  const unsigned long long *pairC; /* 16 words: first BWT position of the suffixes that start with the pair */
  unsigned int numPairSuper;
  unsigned int pairSuperInLds; /* set per launch: the kernel was given numPairSuper * 64 bytes of dynamic LDS for pairSuper32 */
};

/* first entry of level d (the d-letter strings, d >= 1) in the length tables: 4 + 16 + ... + 4^(d-1) */
__host__ __device__ inline unsigned long long awfmLengthTableAt(unsigned d) { return ((1ull << (2u * d)) - 4ull) / 3ull; }

constexpr unsigned kDeepBigShift = 15;
/* exact length of an entry whose 16-bit length field is saturated, from where its range begins */
__device__ __forceinline__ unsigned deepBigLength(const DevIndex &ix, unsigned sp) { return ix.deepBigBySp[sp >> kDeepBigShift]; }
/* format 2 (DevIndex::deepNarrow): the second word of an entry is sp's bits 35..32 | length12 << 4 | next16 << 16 */
constexpr unsigned kDeepWideLengthMask = 0xFFFu, kDeepWideBigShift = 11;
constexpr unsigned kDeepWideMaxBits = 36; /* positions such an entry can name */
__host__ __device__ inline uint2 deepWidePack(unsigned long long sp, unsigned long long length, unsigned next16) {
  const unsigned l12 = length < kDeepWideLengthMask ? (unsigned)length : kDeepWideLengthMask;
  return make_uint2((unsigned)sp, ((unsigned)(sp >> 32) & 0xFu) | (l12 << 4) | (next16 << 16));
}
__device__ __forceinline__ unsigned long long deepWideSp(uint2 e) { return (unsigned long long)e.x | ((unsigned long long)(e.y & 0xFu) << 32); }
/* the bits of an entry's second word that are zero exactly when its range is empty (every format of 8 bytes; uniform) */
__device__ __forceinline__ unsigned deepLengthBits(const DevIndex &ix) {
  return ix.deepNarrow == 2u ? (kDeepWideLengthMask << 4) : (ix.deepNext ? 0xFFFFu : 0xFFFFFFFFu);
}
/* the tables per k-mer length of such an image: sp36 | length28 << 36 */
constexpr unsigned kLengthWideMask = (1u << 28) - 1u, kLengthWideBigShift = 27, kLengthWideBigStride = 512;
__host__ __device__ inline uint2 lengthWidePack(unsigned long long sp, unsigned long long length) {
  const unsigned l28 = length < kLengthWideMask ? (unsigned)length : kLengthWideMask;
  return make_uint2((unsigned)sp, ((unsigned)(sp >> 32) & 0xFu) | (l28 << 4));
}
/* {first position, length} of level `d`'s entry `e` (d >= 1) in either format of the length tables */
__device__ __forceinline__ ulonglong2 lengthEntryOpen(const DevIndex &ix, unsigned d, uint2 e) {
  if (ix.deepNarrow != 2u) return make_ulonglong2((unsigned long long)e.x, (unsigned long long)e.y);
  const unsigned long long sp = deepWideSp(e);
  unsigned long long length = e.y >> 4;
  if (length == kLengthWideMask) length = ix.lengthBig[(d - 1u) * kLengthWideBigStride + (unsigned)(sp >> kLengthWideBigShift)];
  return make_ulonglong2(sp, length);
}
/* The amino alphabet's entries with next-step bits (round 5) are {sp, length12 | next20 << 12}: bit c of next20 says whether
 * the range is still non-empty after one more step with letter c (0..19) -- a hits-only search drops a k-mer whose bit is
 * clear without reading a block: against 2 * 10^9 residues 79 % of random 7-mers occur and 92 % of those die on the next
 * letter.  A length of 0xFFF stands for "4095 or more": the exact one is deepBigBySp[sp >> 11] (two ranges that long begin
 * 4095 or more apart). */
constexpr unsigned kAminoDeepLengthBits = 12, kAminoDeepLengthMask = (1u << kAminoDeepLengthBits) - 1u, kAminoDeepBigShift = kAminoDeepLengthBits - 1u;
/* Amino images that run 64-bit positions (2^32 positions and more; format 2 of DevIndex::deepNarrow, round 6) keep the 8 bytes:
 * sp's bits 35..32 | length8 << 4 | next20 << 12 in the second word -- the next-letter bits where the narrow entries have them --;
 * a length of 0xFF stands for "255 or more", the exact one is the 64-bit word deepBigBySp[sp >> 7] (two ranges that long begin
 * 255 or more apart).  The bits are all set in a table built without them ($AWFM_GPU_DEEP_NEXT=0). */
constexpr unsigned kAminoWideLengthMask = 0xFFu, kAminoWideBigShift = 7;
__host__ __device__ inline uint2 aminoWidePack(unsigned long long sp, unsigned long long length, unsigned next20) {
  const unsigned l8 = length < kAminoWideLengthMask ? (unsigned)length : kAminoWideLengthMask;
  return make_uint2((unsigned)sp, ((unsigned)(sp >> 32) & 0xFu) | (l8 << 4) | (next20 << kAminoDeepLengthBits));
}
__device__ __forceinline__ unsigned long long aminoDeepSp(const DevIndex &ix, uint2 e) { return ix.deepNarrow == 2u ? deepWideSp(e) : (unsigned long long)e.x; }
__device__ __forceinline__ unsigned long long aminoDeepLength(const DevIndex &ix, uint2 e) {
  if (ix.deepNarrow == 2u) { /* uniform */
    const unsigned length = (e.y >> 4) & kAminoWideLengthMask;
    return length == kAminoWideLengthMask ? ((const unsigned long long *)ix.deepBigBySp)[deepWideSp(e) >> kAminoWideBigShift] : (unsigned long long)length;
  }
  if (!ix.deepNext) return e.y;
  const unsigned length = e.y & kAminoDeepLengthMask;
  return length == kAminoDeepLengthMask ? ix.deepBigBySp[e.x >> kAminoDeepBigShift] : length;
}
/* may a k-mer whose next letter (index 0..19; anything else: an ambiguity letter, which has no bit) go on from the entry? */
__device__ __forceinline__ bool aminoDeepNextBit(const DevIndex &ix, uint2 e, unsigned letter) {
  return (ix.deepNarrow != 2u && !ix.deepNext) || letter >= 20u || ((e.y >> (kAminoDeepLengthBits + letter)) & 1u) != 0u;
}
/* {sp, ep} from the two words of an 8-byte entry (format 1 or 2); *next16 (may be NULL): the pair steps that keep the range
 * non-empty (all of them when the table has no such bits) */
__device__ __forceinline__ ulonglong2 deepSeedOpen(const DevIndex &ix, unsigned long long i, uint2 e, unsigned *next16) {
  (void)i;
  if (ix.deepNarrow == 2u) { /* uniform */
    const unsigned long long sp = deepWideSp(e);
    unsigned long long length = (e.y >> 4) & kDeepWideLengthMask;
    if (length == kDeepWideLengthMask) length = ((const unsigned long long *)ix.deepBigBySp)[sp >> kDeepWideBigShift];
    if (next16) *next16 = e.y >> 16;
    return make_ulonglong2(sp, sp + length - 1ull);
  }
  unsigned length = e.y;
  if (ix.deepNext) {
    length = e.y & 0xFFFFu;
    if (length == 0xFFFFu) length = deepBigLength(ix, e.x);
    if (next16) *next16 = e.y >> 16;
  } else if (next16) {
    *next16 = 0xFFFFu;
  }
  return make_ulonglong2((unsigned long long)e.x, (unsigned long long)e.x + length - 1ull);
}
__device__ __forceinline__ ulonglong2 deepSeedEntry(const DevIndex &ix, unsigned long long i) {
  if (ix.deepNarrow) /* image-wide: uniform */
    return deepSeedOpen(ix, i, ((const uint2 *)ix.deepSeed)[i], nullptr);
  return ix.deepSeed[i];
}
/* the same for an amino image (its own entry format when the table has next-step bits) */
__device__ __forceinline__ ulonglong2 aminoDeepSeedEntry(const DevIndex &ix, unsigned long long i) {
  if (ix.deepNarrow) {
    const uint2 e = ((const uint2 *)ix.deepSeed)[i];
    const unsigned long long sp = aminoDeepSp(ix, e);
    return make_ulonglong2(sp, sp + aminoDeepLength(ix, e) - 1ull);
  }
  return ix.deepSeed[i];
}

/* a nucleotide query prepared for the ordered search path (awfm_ordered_kernel.h) */
struct QueryRec {
  unsigned long long codes; /* 2-bit letter codes of the k-mer, last character in bits 1..0 */
  unsigned int index;       /* query number in the batch */
  unsigned int length;      /* characters (1..32), or 0xFFFFFFFF: left to the general kernel */
};

constexpr int kThreads = 256;

/* A launch that carries HIP events on the kernel's own dispatch when it is given any (hipExtLaunchKernelGGL: an event recorded
 * around a kernel is a packet of its own), and the plain launch otherwise: the extended launch with both events null still
 * left the queue idle for 5-7 us in front of the kernel (round 5 traces of a 1.25 * 10^7-k-mer step: the only launches with a
 * gap before them were the extended ones). */
#define AWFM_LAUNCH_WITH_EVENTS(kernel, grid, block, lds, stream, startEvent, stopEvent, ...)                 \
  do {                                                                                                        \
    hipEvent_t start__ = (startEvent), stop__ = (stopEvent);                                                  \
    if (start__ || stop__) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, start__, stop__, 0u, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                   \
  } while (0)

/* Sparse results (awfmGpuSearchHitsCompact): instead of a range / count under every query number, the k-mers with hits
 * are appended to a list {query number, range}, one returning atomic per wave instruction that has any (the order of the
 * list is whatever the waves make it: awfmGpuSortHits puts it in query order).  Entries beyond `cap` are counted, not
 * stored.  count == NULL and kmers != NULL: results IN SEARCH ORDER (awfmGpuSearchHitsInOrder) -- entry q of the order
 * gets {query number, range}, every k-mer one entry, written as whole lines: for batches in which most k-mers have hits,
 * where the stores under the original query numbers are 10^8 partial-line writes.  Both NULL: dense results. */
struct SparseOut {
  unsigned *count;
  unsigned cap;
  unsigned *kmers;
  ulonglong2 *ranges;
};
__device__ __forceinline__ void sparseAppend(const SparseOut &out, bool hit, unsigned index, unsigned long long sp,
                                             unsigned long long ep) {
  const unsigned long long mask = __ballot(hit);
  if (mask == 0ull) return; /* wave-uniform */
  const unsigned lane = threadIdx.x & 63u;
  const int leader = __ffsll((long long)mask) - 1;
  unsigned base = 0;
  if ((int)lane == leader) base = atomicAdd(out.count, (unsigned)__popcll(mask));
  base = (unsigned)__shfl((int)base, leader, 64);
  if (hit) {
    const unsigned slot = base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
    if (slot < out.cap) {
      out.kmers[slot] = index;
      out.ranges[slot] = make_ulonglong2(sp, ep);
    }
  }
}

/* "Lookup first" chosen on the device: sampleAliveKernel leaves the number of its samples that are alive in a device
 * word, and the kernels of BOTH front ends are launched -- the one the sample does not choose returns at once -- so that
 * a search never waits for the host to read that word.  sampleAlive == nullptr: no sample was taken, `otherwise` says. */
__device__ __forceinline__ bool lookupChosen(const unsigned *__restrict__ sampleAlive, const unsigned samples, const bool otherwise) {
  return sampleAlive ? *sampleAlive * 4u < samples : otherwise;
}

/* The full suffix array of an image (AwFmGpuIndex::dDenseSa) as the kernels see it: 32-bit entries, or -- images of 2^32
 * positions and more -- 40-bit entries packed 5 bytes apiece (entry i = bytes 5 i .. 5 i + 4, little endian: the two dwords
 * it lies in are read; one entry in 32 straddles a line).  `wide` is a kernel argument: uniform. */
struct DenseSa {
  const unsigned *words;
  unsigned wide;
};
__device__ __forceinline__ unsigned long long denseSaAt(const DenseSa &sa, unsigned long long i) {
  if (!sa.wide) return (unsigned long long)sa.words[i];
  const unsigned long long byte = i * 5ull, word = byte >> 2;
  const unsigned shift = ((unsigned)byte & 3u) * 8u;
  const unsigned long long both = ((unsigned long long)sa.words[word + 1ull] << 32) | (unsigned long long)sa.words[word];
  return (both >> shift) & 0xFFFFFFFFFFull;
}
inline uint64_t awfmDenseSaBytes(uint64_t n, bool wide) { return wide ? (n + 3ull) / 4ull * 20ull + 16ull : n * 4ull; }
/* n values (32- or 64-bit, below 2^40) into 40-bit entries: a thread takes four values and writes the five dwords they fill */
template <class T>
__global__ void __launch_bounds__(256) packDense40Kernel(const T *__restrict__ in, unsigned long long n, unsigned *__restrict__ out) {
  const unsigned long long quads = (n + 3ull) / 4ull;
  for (unsigned long long q = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; q < quads; q += (unsigned long long)gridDim.x * 256ull) {
    unsigned long long e[4];
    for (unsigned k = 0; k < 4u; k++) e[k] = 4ull * q + k < n ? (unsigned long long)in[4ull * q + k] & 0xFFFFFFFFFFull : 0ull;
    unsigned *w = out + 5ull * q;
    w[0] = (unsigned)e[0];
    w[1] = (unsigned)(e[0] >> 32) | (unsigned)(e[1] << 8);
    w[2] = (unsigned)(e[1] >> 24) | (unsigned)(e[2] << 16);
    w[3] = (unsigned)(e[2] >> 16) | (unsigned)(e[3] << 24);
    w[4] = (unsigned)(e[3] >> 8);
  }
}
/* and back to 32-bit entries (an image below 2^32 positions that is handed a packed array) */
__global__ void __launch_bounds__(256) unpackDense40Kernel(const DenseSa in, unsigned long long n, unsigned *__restrict__ out) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256ull)
    out[i] = (unsigned)denseSaAt(in, i);
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

/* ------------------------------------------------------------------ lane-group helpers */

template <int CTRL>
__device__ __forceinline__ unsigned dppMove(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

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

/* bits 0..(p - 32*slice) of a 32-position slice, clamped: the slice's share of
 * the inclusive prefix mask of ref src/AwFmSimdConfig.c:89-114 */
__device__ __forceinline__ unsigned sliceMask(unsigned p, unsigned slice) {
  int bits = (int)p - (int)(slice * 32) + 1;
  bits = bits < 0 ? 0 : (bits > 32 ? 32 : bits);
  return (unsigned)((1ull << bits) - 1ull);
}

/* position-mask table in LDS: sMask[local * 4 + slice] = bits of slice `slice` at positions <= local */
__device__ __forceinline__ void stageMaskTable(unsigned *sMask) {
  for (unsigned e = threadIdx.x; e < (kBlockMask + 1u) * kSlices; e += blockDim.x) sMask[e] = sliceMask(e >> 2, e & 3u);
}

/* ------------------------------------------------------------------ nucleotide */

/* a 16-byte piece as one 128-bit register tuple */
typedef unsigned Piece __attribute__((ext_vector_type(4)));

/* 1 when (c | 0x20) is one of a,c,g,t,u; branch-free */
__device__ __forceinline__ unsigned nucIsAcgtu(unsigned c) {
  const unsigned d = (c | 0x20u) - 'a';
  return (d < 21u ? 1u : 0u) & (0x180045u >> (d & 31u)); /* bits a=0 c=2 g=6 t=19 u=20 */
}
/* ref src/AwFmLetter.c:4-22: a0 c1 g2 t/u3 '$'5 else 4; branch-free: for a,c,g,t,u the code bits
 * (c>>1)&3 are 0,1,3,2,2 and x^(x>>1) maps them to 0,1,2,3,3 */
__device__ __forceinline__ unsigned nucLetterIndex(unsigned c) {
  const unsigned y = (c >> 1) & 3u;
  const unsigned acgt = y ^ (y >> 1);
  const unsigned other = (c | 0x20u) == '$' ? 5u : 4u;
  return (nucIsAcgtu(c) & 1u) ? acgt : other;
}
/* ref src/AwFmLetter.c:98-125 (tolower(c) is one of acgtu exactly when c|0x20 is) */
__device__ __forceinline__ bool nucIsAmbiguous(unsigned c) { return (nucIsAcgtu(c) & 1u) == 0u; }

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

__device__ __forceinline__ unsigned nucOccSlice(const Piece &pc, const PlaneSel3 &s) {
  return ((pc.x ^ s.x0) | s.d0) & ((pc.y ^ s.x1) | s.d1) & ((pc.z ^ s.x2) | s.d2);
}

/* occurrence bits of letter code (c1,c0) in a nucleotide piece, c0m/c1m = the code bits as all-ones masks.
 * With planes x,y,z: a (00) = y&z, c (01) = x&z, g (10) = x&y, t (11) = x&~y&~z (the literals of ref
 * src/AwFmOccurrence.c:18-31), i.e. (x | a) & ((y ^ t) | c) & ((z ^ t) | g) with one v_bitop3 per factor. */
__device__ __forceinline__ unsigned nucOccFast(const Piece &pc, unsigned c0m, unsigned c1m) {
  const unsigned t0 = __builtin_amdgcn_bitop3_b32(pc.x, c0m, c1m, 0xF1); /* x | (~c0 & ~c1) */
  const unsigned t1 = __builtin_amdgcn_bitop3_b32(pc.y, c0m, c1m, 0x7C); /* (y ^ (c0 & c1)) | (c0 & ~c1) */
  const unsigned t2 = __builtin_amdgcn_bitop3_b32(pc.z, c0m, c1m, 0x7A); /* (z ^ (c0 & c1)) | (~c0 & c1) */
  return t0 & t1 & t2;
}

/* nucleotide superblock bases in LDS: only images of 2^32 or more positions have more than the all-zero entry */
template <bool NARROW>
__device__ __forceinline__ void nucStageSuper(const DevIndex &ix, unsigned long long *sSuper) {
  if (!NARROW)
    for (unsigned e = threadIdx.x; e < kMaxNucSuper * 4u; e += blockDim.x) sSuper[e] = e < ix.numSuper * 4u ? ix.super[e] : 0ull;
}

/*
 * One backward step of a nucleotide query whose next letter is a,c,g or t/u (`letter` 0..3), by the G lanes
 * of its group; lane j holds slices j*S..j*S+S-1 (S = 4/G) of a block.  Everything that does not depend on the
 * block (plane selectors, position masks from the LDS table sMask[local * 4 + slice], C[a]) is computed between
 * issuing the loads and the first use of their data; both blocks are requested before anything waits: the second
 * block's registers start as an "undefined" asm definition and are loaded under the branch, the select happens on
 * the rank results, and an empty asm use keeps every loaded register allocated until the rank is done (a dead
 * component would be re-used for the values computed in the shadow of the load, at the price of a full wait).
 * ref src/AwFmSearch.c:42-159, src/AwFmOccurrence.c:18-31, :170-217.
 */
template <int G, bool NARROW>
__device__ __forceinline__ void nucFastStep(const DevIndex &ix, const unsigned long long *sC, const unsigned long long *sSuper,
                                            const unsigned *sMask, unsigned firstSlice, unsigned letter,
                                            typename PositionType<NARROW>::type &sp,
                                            typename PositionType<NARROW>::type &ep) {
  constexpr int S = (int)kSlices / G;
  typedef typename PositionType<NARROW>::type pos_t;
  const pos_t q0 = sp - 1, q1 = ep;
  const unsigned long long blk0 = q0 >> kBlockShift, blk1 = q1 >> kBlockShift;
  const bool same = blk0 == blk1;
  Piece p0[S], p1[S];
  {
    const Piece *a0 = (const Piece *)(ix.blocks + (blk0 * kSlices + firstSlice));
#pragma unroll
    for (int s = 0; s < S; s++) p0[s] = a0[s];
  }
#pragma unroll
  for (int s = 0; s < S; s++) asm volatile("" : "=v"(p1[s]));
  if (!same) {
    const Piece *a1 = (const Piece *)(ix.blocks + (blk1 * kSlices + firstSlice));
#pragma unroll
    for (int s = 0; s < S; s++) p1[s] = a1[s];
  }
  const unsigned c0m = 0u - (letter & 1u), c1m = 0u - (letter >> 1);
  const unsigned *m0 = sMask + (((unsigned)q0 & kBlockMask) * kSlices + firstSlice);
  const unsigned *m1 = sMask + (((unsigned)q1 & kBlockMask) * kSlices + firstSlice);
  unsigned mask0[S], mask1[S];
#pragma unroll
  for (int s = 0; s < S; s++) {
    mask0[s] = m0[s];
    mask1[s] = m1[s];
  }
  pos_t cLetter = (pos_t)sC[letter];
  pos_t super1 = 0;
  if (!NARROW) { /* 64-bit bases of the superblocks of q0 and q1 (they may differ) */
    const unsigned long long s0 = sSuper[(unsigned)((unsigned long long)q0 >> ix.nucSuperShift) * 4u + letter];
    const unsigned long long s1 = sSuper[(unsigned)((unsigned long long)q1 >> ix.nucSuperShift) * 4u + letter];
    cLetter += (pos_t)s0;
    super1 = (pos_t)(s1 - s0);
  }
  const unsigned sameMask = same ? ~0u : 0u;
  unsigned n0 = 0, n1 = 0;
#pragma unroll
  for (int s = 0; s < S; s++) {
    const unsigned occ0 = nucOccFast(p0[s], c0m, c1m), occ1 = nucOccFast(p1[s], c0m, c1m);
    n0 += __popc(occ0 & mask0[s]);
    n1 += __popc(__builtin_amdgcn_bitop3_b32(occ0, occ1, sameMask, 0xE4) & mask1[s]); /* same ? occ0 : occ1 */
  }
#pragma unroll
  for (int s = 0; s < S; s++) asm volatile("" ::"v"(p0[s]), "v"(p1[s]));
  /* count word of letter a: slice a, i.e. lane a / S, register a % S */
  unsigned c0 = 0, c1 = 0;
#pragma unroll
  for (int s = 0; s < S; s++) {
    c0 = (letter % S) == (unsigned)s ? p0[s].w : c0;
    c1 = (letter % S) == (unsigned)s ? p1[s].w : c1;
  }
  const unsigned base0 = groupShfl<G>(c0, letter / S);
  unsigned base1 = groupShfl<G>(c1, letter / S);
  base1 = same ? base0 : base1;
  const unsigned packed = groupSum<G>(n0 | (n1 << 16));
  sp = cLetter + (pos_t)base0 + (pos_t)(packed & 0xFFFFu);
  ep = cLetter + super1 + (pos_t)base1 + (pos_t)(packed >> 16) - (pos_t)1;
}

/* count of `letter` (0..3, or anything else = X: everything before the block that is not a,c,g,t or '$') before
 * block `blk`, from the count words spread over the group */
template <int G, bool NARROW>
__device__ __forceinline__ typename PositionType<NARROW>::type nucBaseAny(const DevIndex &ix, const unsigned long long *sSuper,
                                                                         const Piece *p, unsigned letter,
                                                                         unsigned long long blk) {
  constexpr int S = (int)kSlices / G;
  typedef typename PositionType<NARROW>::type pos_t;
  const unsigned sb = NARROW ? 0u : (unsigned)(blk >> (ix.nucSuperShift - kBlockShift));
  unsigned mine = 0;
  unsigned long long part = 0;
#pragma unroll
  for (int s = 0; s < S; s++) {
    mine = (letter % S) == (unsigned)s ? p[s].w : mine;
    part += p[s].w;
  }
  if (letter < 4u) {
    const pos_t rel = (pos_t)groupShfl<G>(mine, letter / S);
    return NARROW ? rel : rel + (pos_t)sSuper[sb * 4u + letter];
  }
  unsigned long long acgt = groupSum64<G>(part);
  if (!NARROW) acgt += sSuper[sb * 4u] + sSuper[sb * 4u + 1u] + sSuper[sb * 4u + 2u] + sSuper[sb * 4u + 3u];
  const unsigned long long before = blk << kBlockShift;
  return (pos_t)(before - acgt - (ix.sentinelPos < before ? 1ull : 0ull));
}

/* the rank half of a backward step with any letter index (0..3, 4 = X; ref src/AwFmSearch.c:42-103) from the pieces of
 * the blocks of sp - 1 and ep this lane holds (p1 is ignored when both positions lie in one block) */
template <int G, bool NARROW>
__device__ __forceinline__ void nucStepAnyRank(const DevIndex &ix, const unsigned long long *sC, const unsigned long long *sSuper,
                                               unsigned firstSlice, unsigned letter, const Piece *p0, const Piece *p1,
                                               typename PositionType<NARROW>::type &sp,
                                               typename PositionType<NARROW>::type &ep) {
  constexpr int S = (int)kSlices / G;
  typedef typename PositionType<NARROW>::type pos_t;
  const pos_t q0 = sp - 1, q1 = ep;
  const unsigned long long blk0 = q0 >> kBlockShift, blk1 = q1 >> kBlockShift;
  const bool same = blk0 == blk1;
  const PlaneSel3 sel = nucPlaneSel(letter);
  const unsigned local0 = (unsigned)q0 & kBlockMask, local1 = (unsigned)q1 & kBlockMask;
  const unsigned sameMask = same ? ~0u : 0u;
  unsigned n0 = 0, n1 = 0;
#pragma unroll
  for (int s = 0; s < S; s++) {
    const unsigned occ0 = nucOccSlice(p0[s], sel), occ1 = nucOccSlice(p1[s], sel);
    n0 += __popc(occ0 & sliceMask(local0, firstSlice + s));
    n1 += __popc(__builtin_amdgcn_bitop3_b32(occ0, occ1, sameMask, 0xE4) & sliceMask(local1, firstSlice + s));
  }
  const pos_t base0 = nucBaseAny<G, NARROW>(ix, sSuper, p0, letter, blk0);
  pos_t base1 = nucBaseAny<G, NARROW>(ix, sSuper, p1, letter, blk1);
  base1 = same ? base0 : base1;
  const unsigned packed = groupSum<G>(n0 | (n1 << 16));
  const pos_t cLetter = (pos_t)sC[letter];
  sp = cLetter + base0 + (pos_t)(packed & 0xFFFFu);
  ep = cLetter + base1 + (pos_t)(packed >> 16) - (pos_t)1;
}

/* one backward step with any letter index: loads + rank */
template <int G, bool NARROW>
__device__ __forceinline__ void nucStepAny(const DevIndex &ix, const unsigned long long *sC, const unsigned long long *sSuper,
                                           unsigned firstSlice, unsigned letter,
                                           typename PositionType<NARROW>::type &sp,
                                           typename PositionType<NARROW>::type &ep) {
  constexpr int S = (int)kSlices / G;
  typedef typename PositionType<NARROW>::type pos_t;
  const pos_t q0 = sp - 1, q1 = ep;
  const unsigned long long blk0 = q0 >> kBlockShift, blk1 = q1 >> kBlockShift;
  const bool same = blk0 == blk1;
  Piece p0[S], p1[S];
  {
    const Piece *a0 = (const Piece *)(ix.blocks + (blk0 * kSlices + firstSlice));
#pragma unroll
    for (int s = 0; s < S; s++) p0[s] = a0[s];
  }
#pragma unroll
  for (int s = 0; s < S; s++) asm volatile("" : "=v"(p1[s]));
  if (!same) {
    const Piece *a1 = (const Piece *)(ix.blocks + (blk1 * kSlices + firstSlice));
#pragma unroll
    for (int s = 0; s < S; s++) p1[s] = a1[s];
  }
#pragma unroll
  for (int s = 0; s < S; s++) asm volatile("" ::"v"(p0[s]), "v"(p1[s]));
  nucStepAnyRank<G, NARROW>(ix, sC, sSuper, firstSlice, letter, p0, p1, sp, ep);
}

/* ------------------------------------------------------------------ amino */

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

__device__ __forceinline__ void aminoStageTables(AminoShared &t) {
  if (threadIdx.x < 32) {
    t.letterOfAscii[threadIdx.x] = kAminoTables.letterOfAscii[threadIdx.x];
    t.letterOfCode[threadIdx.x] = kAminoTables.letterOfCode[threadIdx.x];
    if (threadIdx.x < 24) t.planeMask[threadIdx.x] = kAminoTables.planeMask[threadIdx.x];
  }
}

__device__ __forceinline__ unsigned aminoLetterIndex(const AminoShared &t, unsigned c) {
  return c == '$' ? 21u : (unsigned)t.letterOfAscii[c & 31u];
}
__device__ __forceinline__ bool aminoIsAmbiguous(unsigned c) {
  const unsigned l = c | 0x20u; /* equals tolower(c) whenever the result is z, x or b */
  return ((l == 'z') | (l == 'x') | (l == 'b')) != 0;
}

__device__ __forceinline__ unsigned aminoLiteral(unsigned plane, unsigned ones, unsigned zeros, unsigned j) {
  const unsigned x = 0u - ((zeros >> j) & 1u);
  const unsigned d = (((ones | zeros) >> j) & 1u) - 1u;
  return (plane ^ x) | d;
}

/* occurrence bits of a letter in one amino slice: lo = {b0,b1,b2,b3}, hi = {b4, c01, c23, c45} */
__device__ __forceinline__ unsigned aminoOccSlice(const Piece &lo, const Piece &hi, unsigned ones, unsigned zeros) {
  return aminoLiteral(lo.x, ones, zeros, 0) & aminoLiteral(lo.y, ones, zeros, 1) &
         aminoLiteral(lo.z, ones, zeros, 2) & aminoLiteral(lo.w, ones, zeros, 3) &
         aminoLiteral(hi.x, ones, zeros, 4);
}

/* 16-bit count `sub` (0..5) of a slice's second piece, picked with shifts (a select chain on vector components
 * indexed dynamically makes hipcc move the vector to scratch) */
__device__ __forceinline__ unsigned aminoCount16(const Piece &hi, unsigned sub) {
  const unsigned long long c03 = ((unsigned long long)hi.z << 32) | hi.y;
  const unsigned long long c45 = hi.w;
  return (unsigned)((sub >= 4u ? c45 : c03) >> (16u * (sub & 3u))) & 0xFFFFu;
}

/* one backward step of an amino query by the G lanes of its group (G = 4 or 2); lane j holds slices j*S..j*S+S-1,
 * two pieces each.  ref src/AwFmSearch.c:105-159, src/AwFmOccurrence.c:52-135.  The superblock bases are read
 * beside the blocks (one address for the lanes of a group); loads and rank are arranged as in nucFastStep. */
template <int G, bool NARROW>
__device__ __forceinline__ void aminoStepAny(const DevIndex &ix, const unsigned long long *sC, const AminoShared &t,
                                             const unsigned *sMask, unsigned firstSlice, unsigned letter,
                                             typename PositionType<NARROW>::type &sp,
                                             typename PositionType<NARROW>::type &ep) {
  constexpr int S = (int)kSlices / G;
  typedef typename PositionType<NARROW>::type pos_t;
  const pos_t q0 = sp - 1, q1 = ep;
  const unsigned long long blk0 = q0 >> kBlockShift, blk1 = q1 >> kBlockShift;
  const bool same = blk0 == blk1;
  const unsigned safe = letter < 24u ? letter : 23u;
  Piece p0[S][2], p1[S][2];
  {
    const Piece *a0 = (const Piece *)(ix.blocks + (blk0 * kSlices + firstSlice) * 2ull);
#pragma unroll
    for (int s = 0; s < S; s++) {
      p0[s][0] = a0[2 * s];
      p0[s][1] = a0[2 * s + 1];
    }
  }
  const unsigned long long super0 = ix.super[(unsigned long long)(q0 >> kAminoSuperShift) * kAminoSuperStride + safe];
#pragma unroll
  for (int s = 0; s < S; s++) asm volatile("" : "=v"(p1[s][0]), "=v"(p1[s][1]));
  unsigned long long super1;
  asm volatile("" : "=v"(super1));
  if (!same) {
    const Piece *a1 = (const Piece *)(ix.blocks + (blk1 * kSlices + firstSlice) * 2ull);
#pragma unroll
    for (int s = 0; s < S; s++) {
      p1[s][0] = a1[2 * s];
      p1[s][1] = a1[2 * s + 1];
    }
    super1 = ix.super[(unsigned long long)(q1 >> kAminoSuperShift) * kAminoSuperStride + safe];
  }
  const unsigned pm = t.planeMask[safe];
  const unsigned ones = pm & 0xFFu, zeros = pm >> 8;
  const unsigned *m0 = sMask + (((unsigned)q0 & kBlockMask) * kSlices + firstSlice);
  const unsigned *m1 = sMask + (((unsigned)q1 & kBlockMask) * kSlices + firstSlice);
  unsigned mask0[S], mask1[S];
#pragma unroll
  for (int s = 0; s < S; s++) {
    mask0[s] = m0[s];
    mask1[s] = m1[s];
  }
  const pos_t cLetter = (pos_t)sC[safe];
  const unsigned sameMask = same ? ~0u : 0u;
  const unsigned slice = safe / 6u, sub = safe % 6u; /* where the letter's 16-bit count lives */
  unsigned n0 = 0, n1 = 0, c0 = 0, c1 = 0;
#pragma unroll
  for (int s = 0; s < S; s++) {
    const unsigned occ0 = aminoOccSlice(p0[s][0], p0[s][1], ones, zeros), occ1 = aminoOccSlice(p1[s][0], p1[s][1], ones, zeros);
    n0 += __popc(occ0 & mask0[s]);
    n1 += __popc(__builtin_amdgcn_bitop3_b32(occ0, occ1, sameMask, 0xE4) & mask1[s]);
    const unsigned w0 = aminoCount16(p0[s][1], sub), w1 = aminoCount16(p1[s][1], sub);
    c0 = (slice % S) == (unsigned)s ? w0 : c0;
    c1 = (slice % S) == (unsigned)s ? w1 : c1;
  }
#pragma unroll
  for (int s = 0; s < S; s++) asm volatile("" ::"v"(p0[s][0]), "v"(p0[s][1]), "v"(p1[s][0]), "v"(p1[s][1]));
  asm volatile("" ::"v"(super1));
  const pos_t base0 = (pos_t)super0 + (pos_t)groupShfl<G>(c0, slice / S);
  pos_t base1 = (pos_t)super1 + (pos_t)groupShfl<G>(c1, slice / S);
  base1 = same ? base0 : base1;
  const unsigned packed = groupSum<G>(n0 | (n1 << 16));
  sp = cLetter + base0 + (pos_t)(packed & 0xFFFFu);
  ep = cLetter + base1 + (pos_t)(packed >> 16) - (pos_t)1;
}

/* ------------------------------------------------------------------ image build kernels */

/* absolute base counts at superblock starts, from the reference-layout blocks (ref src/AwFmIndex.h:55-65: 160-B
 * blocks = planes [3][4] + counts [8] as 64-bit words; 352-B blocks = planes [5][4] + counts [24]) */
__global__ void gatherSuperKernel(const unsigned long long *__restrict__ ref, unsigned long long numRefBlocks, int amino,
                                  unsigned superShift, unsigned numSuper, unsigned long long *__restrict__ super) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned stride = amino ? kAminoSuperStride : 4u;
  if (t >= numSuper * stride) return;
  const unsigned sb = t / stride, a = t % stride;
  const unsigned long long refBlk = ((unsigned long long)sb << superShift) >> 8;
  unsigned long long v = 0;
  if (refBlk < numRefBlocks && a < (amino ? 21u : 4u)) v = amino ? ref[refBlk * 44ull + 20u + a] : ref[refBlk * 20ull + 12u + a];
  super[t] = v;
}

/* reference-layout blocks -> device layout; also finds the sentinel's BWT position.  One thread per slice. */
__global__ void relayoutNucKernel(const unsigned long long *__restrict__ ref, unsigned long long numRefBlocks,
                                  unsigned long long bwtLength, unsigned superShift,
                                  const unsigned long long *__restrict__ super, uint4 *__restrict__ out,
                                  unsigned long long *__restrict__ sentinelPos) {
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long blk = t >> 2; /* device block */
  const unsigned k = (unsigned)t & 3u;   /* slice = letter whose count this piece carries */
  const unsigned long long refBlk = blk >> 1;
  if (refBlk >= numRefBlocks) return;
  const unsigned half = (unsigned)blk & 1u;
  const unsigned long long *src = ref + refBlk * 20ull;
  auto planeWord = [&](unsigned plane, unsigned w) -> unsigned { /* 32-bit word w (0..7) of a 256-bit plane */
    return (unsigned)(src[4u * plane + (w >> 1)] >> ((w & 1u) * 32u));
  };
  const unsigned w = 4u * half + k;
  const unsigned b0 = planeWord(0, w), b1 = planeWord(1, w), b2 = planeWord(2, w);
  unsigned long long count = src[12u + k]; /* letter k before the 256-position block */
  if (half) {
    const PlaneSel3 sel = nucPlaneSel(k);
    for (unsigned j = 0; j < 4; j++) {
      Piece pc;
      pc.x = planeWord(0, j);
      pc.y = planeWord(1, j);
      pc.z = planeWord(2, j);
      pc.w = 0;
      count += __popc(nucOccSlice(pc, sel));
    }
  }
  count -= super[(blk >> (superShift - kBlockShift)) * 4ull + k];
  out[blk * kSlices + k] = make_uint4(b0, b1, b2, (unsigned)count);
  const unsigned sentinelBits = b2 & ~b1 & ~b0; /* code 100b, ref src/AwFmLetter.c:44-47 */
  if (sentinelBits) {
    const unsigned long long pos = (blk << kBlockShift) + k * 32u + (unsigned)(__ffs((int)sentinelBits) - 1);
    if (pos < bwtLength) *sentinelPos = pos;
  }
}

__global__ void relayoutAminoKernel(const unsigned long long *__restrict__ ref, unsigned long long numRefBlocks,
                                    unsigned long long bwtLength, const unsigned long long *__restrict__ super,
                                    uint4 *__restrict__ out, unsigned long long *__restrict__ sentinelPos) {
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long blk = t >> 2;
  const unsigned k = (unsigned)t & 3u;
  const unsigned long long refBlk = blk >> 1;
  if (refBlk >= numRefBlocks) return;
  const unsigned half = (unsigned)blk & 1u;
  const unsigned long long *src = ref + refBlk * 44ull;
  auto planeWord = [&](unsigned plane, unsigned w) -> unsigned {
    return (unsigned)(src[4u * plane + (w >> 1)] >> ((w & 1u) * 32u));
  };
  unsigned b[5];
  for (unsigned j = 0; j < 5; j++) b[j] = planeWord(j, 4u * half + k);
  unsigned c16[6];
  for (unsigned s = 0; s < 6; s++) {
    const unsigned letter = 6u * k + s;
    unsigned long long count = 0;
    if (letter < 21u) {
      count = src[20u + letter];
      if (half) {
        const unsigned pm = kAminoTables.planeMask[letter];
        for (unsigned j = 0; j < 4; j++) {
          Piece lo, hi;
          lo.x = planeWord(0, j);
          lo.y = planeWord(1, j);
          lo.z = planeWord(2, j);
          lo.w = planeWord(3, j);
          hi.x = planeWord(4, j);
          hi.y = hi.z = hi.w = 0;
          count += __popc(aminoOccSlice(lo, hi, pm & 0xFFu, pm >> 8));
        }
      }
      count -= super[(blk >> (kAminoSuperShift - kBlockShift)) * kAminoSuperStride + letter];
    }
    c16[s] = (unsigned)count & 0xFFFFu;
  }
  out[(blk * kSlices + k) * 2ull] = make_uint4(b[0], b[1], b[2], b[3]);
  out[(blk * kSlices + k) * 2ull + 1ull] = make_uint4(b[4], c16[0] | (c16[1] << 16), c16[2] | (c16[3] << 16), c16[4] | (c16[5] << 16));
  unsigned sentinelBits = ~(b[0] | b[1] | b[2] | b[3] | b[4]); /* code 00000 */
  while (sentinelBits) {
    const unsigned bit = (unsigned)(__ffs((int)sentinelBits) - 1);
    sentinelBits &= sentinelBits - 1u;
    const unsigned long long pos = (blk << kBlockShift) + k * 32u + bit;
    if (pos < bwtLength) *sentinelPos = pos;
  }
}

}  // namespace

struct AwFmGpuIndex {
  /* a lane: a second host-side handle on the device image of `shares` (same device buffers, own staging
   * buffers and locks), so that two host threads can overlap their pack/scatter with each other's transfers
   * and kernels without a second copy of the index; owns none of the device arrays */
  AwFmGpuIndex *shares = nullptr;
  int device = 0;
  bool amino = false;
  DevIndex dev{};
  void *dBlocks = nullptr;
  void *dSuper = nullptr;
  void *dSeed = nullptr;
  void *dSa = nullptr;
  void *dPrefix = nullptr;
  void *dDeepSeed = nullptr;
  uint64_t deepSeedBytes = 0;
  double deepSeedBuildSeconds = 0.0;    /* wall time of the last construction of the deeper table (reporting) */
  double deepSeedAllocSeconds = 0.0;    /* ... of which inside hipMalloc */
  uint64_t deepSeedTransientBytes = 0;  /* device memory that construction held beyond the table itself, at its peak */
  void *dDeepBig = nullptr; /* the deeper table's lengths of 65535 and more, by where the range begins (DevIndex::deepBigBySp) */
  /* optional tables of the k-mer lengths below the deeper table's (awfmGpuBuildLengthTables), built by the first
   * mixed-length batch that can use them; owned by the primary, found there by its lanes (under lengthMutex) */
  std::mutex lengthMutex;
  void *dLengthTable = nullptr;
  unsigned lengthDepths = 0; /* levels 1 .. lengthDepths */
  bool lengthTried = false;  /* a construction was attempted and failed, or is done */
  unsigned lengthRetryIn = 0; /* calls since it failed: every 64th tries again */
  void *dLengthBig = nullptr; /* DevIndex::lengthBig of the tables in the wide format (set before dLengthTable is published) */
  uint64_t lengthTableBytes = 0;
  double lengthTableBuildSeconds = 0.0;
  void *dDenseSa = nullptr; /* optional full suffix array: 32-bit entries, or (denseWide) 40-bit ones packed 5 bytes apiece */
  bool denseWide = false;
  uint64_t denseSaBytes = 0;
  double denseSaBuildSeconds = 0.0; /* wall time of the automatic construction (reporting) */
  void *dPairBlocks = nullptr, *dPairSuper = nullptr, *dPairSuper32 = nullptr, *dPairC = nullptr; /* pair image */
  uint64_t pairBytes = 0;
  /* what became of the optional accelerators when the image was made -- which ones it would have got by its size and did not,
   * and why (awfmGpuIndexDescribe; the searches simply run without them) */
  std::string accelNotes;
  uint64_t deviceBytes = 0;
  uint64_t numBlocks = 0; /* device blocks (128 positions each) */
  AwFmGpuKernel kernel = AWFM_GPU_KERNEL_AUTO;
  /* testing: run the 64-bit-position kernels although bwtLength < 2^32 (awfmGpuIndexSetWide, $AWFM_GPU_FORCE_WIDE) */
  bool forceWide = false;
  int numCUs = 256;
  /* grow-only workspace for the host-buffer entry points */
  std::mutex workMutex;
  void *dWork = nullptr;
  size_t workBytes = 0;
  void *dHits = nullptr; /* positions of the host-buffer locate calls, grow-only like dWork */
  size_t hitsBytes = 0;
  mutable uint64_t hitBudgetAuto = 0; /* awfmGpuHitBudget's automatic value, asked of the device once (0: not yet) */
  void *dSparse = nullptr; /* temporaries of awfmGpuSortHits, grow-only (under orderMutex) */
  size_t sparseBytes = 0;
  hipEvent_t windowEvent[2] = {nullptr, nullptr}; /* the two hit windows in flight of awfmGpuLocateHostWindows */
  /* ordered search path (awfm_gpu_ordered.hip): grow-only scratch shared by all searches on this image;
   * the event orders its re-use across streams */
  int orderMode = -1; /* -1 auto, 0 off, 1 on */
  std::mutex orderMutex;
  /* who used a piece of scratch last, so that the next user on ANOTHER stream waits for it (and one on the same stream,
   * which is ordered behind it anyway, does not pay for an event) */
  struct StreamGate {
    hipEvent_t done = nullptr;
    bool recorded = false; /* `done` says when the last use is over */
    bool pending = false;  /* the last use left no event: it is recorded on lastStream when another stream needs it */
    hipStream_t lastStream = nullptr;
    std::thread::id lastThread; /* hipStreamPerThread is one handle for a different stream in every thread */
  };
  /* the scratch of a seed-order search: kOrderSlots of them, so that two callers on two streams overlap instead of queueing
   * (a caller keeps the slot it used last; a newcomer takes the one that has rested longest) */
  struct OrderSlot {
    void *mem = nullptr;
    size_t bytes = 0;
    StreamGate gate;
    unsigned long long lastUse = 0;
    /* lookupPrepKernel's two sample words (awfm_ordered_kernel.h): which of them the last search left zero for the next
     * one, or -1 when another kind of search has used the counter block since (the words are then zeroed by a memset) */
    int prepParity = -1;
  };
  static constexpr int kOrderSlots = 2;
  OrderSlot orderSlot[kOrderSlots];
  unsigned long long orderUses = 0;
  int orderCur = 0;           /* the slot of the search being enqueued (under orderMutex) */
  int orderPrevParity = -1;   /* that slot's prepParity as the search before left it (orderBeginSlot resets the slot's own) */
  void *dOrder = nullptr;     /* = orderSlot[orderCur].mem */
  size_t orderBytes = 0;      /* = orderSlot[orderCur].bytes */
  hipEvent_t orderDoneEvent = nullptr; /* set while a search is enqueued: its last kernel carries it (stop event) ... */
  bool orderDoneArmed = false;         /* ... and says so here */
  StreamGate sparseGate; /* dSparse */
  /* $AWFM_GPU_TIME_ORDERED: [0],[1] around the front end that looks the table up (encodeLookupKernel), [2],[3] around
   * orderedSearchKernel; which pair is the call's dominant kernel follows from orderLookup */
  hipEvent_t orderTiming[4] = {nullptr, nullptr, nullptr, nullptr}; /* the current entry of orderLog */
  bool orderTimedFront = false, orderTimedKernel = false;
  /* one entry per timed search, so that a caller can time every step of a loop without waiting inside it
   * (awfmGpuOrderedKernelLog): a ring of kOrderLogMax entries, events created on first use */
  struct OrderLogEntry {
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool front = false, kernel = false;
  };
  static constexpr unsigned kOrderLogMax = 1024;
  std::vector<OrderLogEntry> orderLog;
  unsigned long long orderLogCount = 0; /* searches timed since the log was last read */
  /* did the last bucketed search keep its k-mers by encodeLookupKernel?  0 no, 1 yes, 2 decided ON THE DEVICE from the
   * sample (no host wait inside the search): the sample's count is the device word orderSampleAt, the pass was taken when
   * 4 x that count < orderSamples */
  int orderLookup = 0;
  const unsigned *orderSampleAt = nullptr;
  unsigned orderSamples = 0;
  bool orderLookupFused = false;               /* ... by lookupSearchKernel, which searched the k-mers it kept itself */
  const unsigned *orderFusedKeptAt = nullptr;  /* its survivor counters (kFusedCounters words, 64 B apart) */
  const unsigned *orderKeptAt = nullptr; /* device word: k-mers that search ordered (after encodeLookupKernel: the ones it kept) */
  /* Lookup prediction (round 5, awfm_gpu_ordered.hip): a sampled search publishes {its number, the sample's count} in
   * page-locked host memory when its sample is in; a later search of the same k-mer length reads the newest verdict -- no
   * wait: whatever has arrived -- and launches only the front end it names (the lookup kernel with what it cannot finish
   * left to the general kernel, or the ordering passes and the ordered kernel), both when there is none.  Either front end
   * alone is correct for any batch; a verdict that contradicts the mode its own search ran in switches prediction off for
   * the next kPredictHoldoff searches.  Under orderMutex. */
  struct LookupPredict {
    /* (tag << 32) | k-mers of the sample still alive; tag = the search's number (22 bits, never 0) | the front end(s) it
     * launched << 22 (0 both, 1 lookup only, 2 ordered only) | its k-mer length << 24 */
    unsigned long long *verdictHost = nullptr;
    unsigned searches = 0; /* the number of the last sampled search */
    int lastFront = -1;    /* what it launched (reporting) */
    unsigned lastJudged = 0; /* the newest verdict that was compared with its own search's mode */
    unsigned holdoff = 0;      /* searches that still launch both front ends after a miss */
    unsigned holdoffNext = 8;  /* what the next miss sets it to (doubles per miss up to 1024, back to 8 after 64 good predictions) */
    unsigned agreed = 0;
  } predict;
  int lastSearchExact = 0; /* awfmGpuLastSearchWasExactLookup */
  /* Accelerators built BEHIND the first searches (round 6).  An image made by awfmGpuIndexAcquireAll -- the drop-in entry
   * points' way -- is usable as soon as its blocks, seed table, sampled array and pair image are on the device (the reference's
   * index is usable the moment awFmReadIndexFromFile returns, ref src/AwFmFile.c:195-449); its deeper table and full suffix array
   * are made by a thread of their own on a stream of their own and installed between two calls (awfmGpuAdoptAccelerators).
   * Searches give the reference's results with or without them.  accelState: 0 nothing pending, 1 being built, 2 built and
   * waiting to be installed. */
  std::thread accelThread;
  std::atomic<int> accelState{0};
  struct PendingAccel {
    void *deepTable = nullptr, *deepBig = nullptr;
    uint64_t deepBytes = 0, deepBigBytes = 0, deepTransient = 0;
    unsigned deepK = 0, deepFormat = 0, deepNext = 0, numDeepBig = 0;
    double deepSeconds = 0.0, deepAllocSeconds = 0.0;
    void *dense = nullptr;
    bool denseWide = false;
    uint64_t denseBytes = 0;
    double denseSeconds = 0.0;
    std::string notes;
  } pendingAccel;
  std::mutex aosMutex;       /* serialises the AoS entry points (they share the pinned buffers) */
  void *pinned[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t pinnedBytes[4] = {0, 0, 0, 0};
  /* chunked host-buffer pipeline (awfm_gpu_stream.hip): three slots of device buffers + page-locked staging,
   * created on first use, one batch at a time */
  std::mutex streamMutex;
  struct AwFmGpuStreamState *streamState = nullptr;
};

/* Test and diagnostics hooks -- none of them selects a faster path --, all behind ONE environment variable:
 * $AWFM_GPU_DIAG = "key=value,key=value,..." (include/awfm_gpu.h lists the keys).  Returns the value of `key` (up to 31
 * characters, in a buffer of the calling thread), or NULL. */
inline const char *awfmGpuDiag(const char *key) {
  const char *env = awfmKnob(AWFM_KNOB_DIAG);
  if (!env) return nullptr;
  static thread_local char value[32];
  const size_t keyLen = strlen(key);
  for (const char *at = env; *at;) {
    const char *end = strchr(at, ',');
    const size_t len = end ? (size_t)(end - at) : strlen(at);
    if (len > keyLen && !strncmp(at, key, keyLen) && at[keyLen] == '=') {
      const size_t n = len - keyLen - 1 < sizeof(value) - 1 ? len - keyLen - 1 : sizeof(value) - 1;
      memcpy(value, at + keyLen + 1, n);
      value[n] = 0;
      return value;
    }
    at += len + (end ? 1 : 0);
  }
  return nullptr;
}

/* sizes of the device block array and of the superblock table of an index */
inline uint64_t awfmDeviceBlocks(uint64_t bwtLength) { return 2 * awfmNumBlocks(bwtLength); }
inline uint64_t awfmDeviceBlockBytes(bool amino) { return amino ? 128 : 64; }
/* log2 of the positions per superblock.  Nucleotide: 32.  $AWFM_GPU_DIAG nuc_super_shift = 13..31, or "auto" (the
 * smallest shift >= 13 that gives at most 48 superblocks), makes them smaller so that the parity tests reach the
 * several-superblock arithmetic an index of 2^32 or more positions runs -- such an image always uses the 64-bit
 * kernels, the only ones that read the superblock bases. */
inline unsigned awfmSuperShift(bool amino, uint64_t bwtLength) {
  if (amino) return kAminoSuperShift;
  if (const char *env = awfmGpuDiag("nuc_super_shift")) {
    if (!strcmp(env, "auto")) {
      unsigned shift = 13;
      while (((bwtLength - 1) >> shift) + 1 > 48) shift++;
      return shift < kNucSuperShift ? shift : kNucSuperShift;
    }
    const int v = atoi(env);
    if (v >= 13 && v < (int)kNucSuperShift) return (unsigned)v;
  }
  return kNucSuperShift;
}
inline uint64_t awfmNumSuper(uint64_t bwtLength, bool amino, unsigned superShift) {
  (void)amino;
  return ((bwtLength - 1) >> superShift) + 1;
}
inline uint64_t awfmSuperBytes(uint64_t bwtLength, bool amino, unsigned superShift) {
  return awfmNumSuper(bwtLength, amino, superShift) * (amino ? kAminoSuperStride : 4u) * 8u;
}

/* 32-bit BWT positions in the kernels: exact whenever bwtLength < 2^32 (ref src/AwFmIndex.h:88-91 is 64-bit
 * throughout; the NARROW = false instantiations are that arithmetic) */
inline bool awfmImageNarrow(const AwFmGpuIndex *g) {
  return !g->forceWide && g->dev.bwtLength < (1ull << 32) && (g->amino || g->dev.numSuper == 1);
}

/* where the kernels that use the pair image keep its 32-bit superblock bases (64 B per 2^23 positions): in dynamic LDS,
 * or read from memory beside the blocks */
inline bool awfmPairSuperInLds(const AwFmGpuIndex *g) {
  return g->dev.numPairSuper * (kPairSuperStride * 4u) <= 32768u; /* measured: 4.46 ms from LDS against 4.91 ms from memory (10^8 random 21-mers) */
}

/* RAII hipSetDevice */
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

/* ---- shared by the translation units of the shim (awfm_gpu*.hip) ---- */
/* The stream the image-construction code of the CALLING THREAD launches on: the null stream (everything an image is made of is
 * then complete when the call that makes it returns), or -- the thread that builds an image's deeper table and full suffix
 * array BEHIND the first searches (round 6: awfm_gpu_image.hip, awfmGpuIndexCreate) -- a non-blocking stream of its own, so
 * that its kernels neither wait for the searches' streams nor make them wait. */
extern thread_local hipStream_t awfmGpuSetupStream;
inline hipError_t awfmGpuSetupSync() { return awfmGpuSetupStream ? hipStreamSynchronize(awfmGpuSetupStream) : hipDeviceSynchronize(); }
inline hipError_t awfmGpuSetupMemset(void *p, int value, size_t bytes) { return hipMemsetAsync(p, value, bytes, awfmGpuSetupStream); }
inline hipError_t awfmGpuSetupToHost(void *dst, const void *src, size_t bytes) {
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, awfmGpuSetupStream);
  return e != hipSuccess ? e : hipStreamSynchronize(awfmGpuSetupStream);
}
/* Persistent grid: the kernels stride over the work, so the grid is exactly what is resident (blocksPerCU from the occupancy
 * query for that kernel); a larger grid would run as a second, under-filled round. */
template <class Kernel>
unsigned gridFor(uint64_t groups, const AwFmGpuIndex *g, Kernel kernel, unsigned groupsPerBlock, size_t dynamicLds = 0, int threads = kThreads) {
  int perCU = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kernel, threads, dynamicLds) != hipSuccess || perCU < 1) perCU = 4;
  if (perCU > 8) perCU = 8;
  const uint64_t blocks = (groups + groupsPerBlock - 1) / groupsPerBlock;
  const uint64_t cap = (uint64_t)g->numCUs * (uint64_t)perCU;
  return (unsigned)(blocks < cap ? (blocks ? blocks : 1) : cap);
}
inline size_t alignUp(size_t v, size_t a) { return (v + a - 1) / a * a; }
/* the image's full suffix array as a kernel argument */
inline DenseSa denseSaOf(const AwFmGpuIndex *g) {
  DenseSa sa;
  sa.words = (const unsigned *)g->dDenseSa;
  sa.wide = g->denseWide ? 1u : 0u;
  return sa;
}
/* awfm_gpu_image.hip: the image's grow-only work buffer (the caller holds workMutex); lanes that cooperate on one k-mer in the
 * general kernel and the walk; the lanes of a primary image */
enum AwFmReturnCode awfmGpuEnsureWork(AwFmGpuIndex *g, size_t bytes);
int awfmGpuLanesPerQuery(const AwFmGpuIndex *g);
std::vector<AwFmGpuIndex *> awfmGpuLanesOf(const AwFmGpuIndex *primary);
/* holds the work and AoS locks of every lane of a primary image for the lifetime of the object */
struct AwFmGpuLaneLocks {
  std::vector<AwFmGpuIndex *> lanes;
  explicit AwFmGpuLaneLocks(const AwFmGpuIndex *primary) : lanes(awfmGpuLanesOf(primary)) {
    for (AwFmGpuIndex *lane : lanes) {
      lane->aosMutex.lock();
      lane->workMutex.lock();
    }
  }
  ~AwFmGpuLaneLocks() {
    for (AwFmGpuIndex *lane : lanes) {
      lane->workMutex.unlock();
      lane->aosMutex.unlock();
    }
  }
};
/* awfm_gpu_locate.hip: LF-walk + sampled-SA kernels over `totalHits` BWT positions stored in dPositions (in place, or to `out`);
 * stepCap != 0: the construction of the full suffix array (walks given up after so many steps are parked) */
enum AwFmReturnCode awfmGpuLaunchLocate(AwFmGpuIndex *g, unsigned long long totalHits, unsigned long long *dPositions, hipStream_t s,
                                        unsigned long long *out = nullptr, const unsigned long long *totalOnDevice = nullptr, unsigned stepCap = 0u);
/* awfm_gpu_dense_sa.hip: the full suffix array of an image that was just created or adopted ($AWFM_GPU_DENSE_SA, else by its size) */
enum AwFmReturnCode awfmGpuApplyDenseSaAuto(AwFmGpuIndex *g);
/* ... the same decision and construction without touching the image (the thread that builds behind the first searches): the
 * array, its entry width and size, the seconds it took; notes: what was not built and why */
enum AwFmReturnCode awfmGpuBuildDenseSaAuto(const AwFmGpuIndex *g, void **arrayOut, bool *wideOut, uint64_t *bytesOut, double *secondsOut,
                                            std::string *notes);
/* awfm_gpu_image.hip: installs what an image's builder thread has finished.  wait: join the thread first (the explicit
 * entry points); otherwise only when it is done and the image's locks are free right now.  lanes: the image's lanes, listed by
 * a caller that holds the registry's lock (NULL: looked up here) */
void awfmGpuAdoptAccelerators(AwFmGpuIndex *g, bool wait, const std::vector<AwFmGpuIndex *> *lanes = nullptr);

/* blocks + superblock table of the device image from reference-layout blocks already on the device (current device,
 * null stream).  dBlocks / dSuper are allocated by the caller: awfmDeviceBlocks x awfmDeviceBlockBytes, awfmSuperBytes.
 * Synchronous; false with awfmGpuLastError set on failure. */
bool awfmGpuRelayout(const void *dRefBlocks, uint64_t bwtLength, bool amino, unsigned superShift, void *dBlocks,
                     void *dSuper, unsigned long long *sentinelPosOut);

/* ordered hits-only search; 1 = searched, 0 = does not apply, <0 = -AwFmReturnCode (awfm_gpu_ordered.hip) */
/* packed: dChars is one 64-bit word per fixed-length k-mer (2-bit codes) instead of ASCII */
int awfmGpuOrderedSearch(AwFmGpuIndex *g, hipStream_t s, const uint8_t *dChars, const unsigned long long *off,
                         uint32_t fixedLength, unsigned long long nq, ulonglong2 *rng, uint32_t *dCounts, bool packed = false,
                         bool rangesOfHitsOnly = false);

/* hits-only search of a large fixed-length amino batch through the deeper table (awfm_amino_lookup_kernel.h):
 * 1 = searched, 0 = does not apply, < 0 = -AwFmReturnCode */
int awfmGpuAminoLookupSearch(AwFmGpuIndex *g, hipStream_t s, const uint8_t *dChars, uint32_t fixedLength, unsigned long long nq,
                             ulonglong2 *rng, uint32_t *dCounts, bool rangesOfHitsOnly);

/* adopts device buffers that already hold a complete image (used by the GPU builder) */
AwFmGpuIndex *awfmGpuIndexAdopt(const struct AwFmIndex *index, int device, void *dBlocks, void *dSuper, unsigned superShift,
                                void *dSeed, void *dSa, void *dPrefix, unsigned long long sentinelPos, uint64_t deviceBytes);
void awfmGpuIndexRegister(const struct AwFmIndex *index, AwFmGpuIndex *g);
/* awfm_gpu_pair.hip: builds (enable) or drops the pair image of a primary nucleotide image; the caller holds whatever
 * locks the image needs and copies the view into its lanes */
enum AwFmReturnCode awfmGpuApplyPairImage(AwFmGpuIndex *g, bool enable);
/* awfm_gpu_stream.hip: drops the pipeline slots of an image (called by awfmGpuIndexDestroy) */
void awfmGpuStreamStateFree(AwFmGpuIndex *g);
/* hit offsets of a chunk without a host wait: exclusive scan of the 32-bit counts (dCounts != NULL) or of the range
 * lengths into dHitOffsets[0..n], the total also copied to *pinnedTotal (page-locked) on the stream */
enum AwFmReturnCode awfmGpuHitOffsetsAsync(AwFmGpuIndex *g, const uint32_t *dCounts, const struct AwFmSearchRange *dRanges,
                                           uint64_t numQueries, uint64_t *dHitOffsets, void *dScratch,
                                           unsigned long long *pinnedTotal, hipStream_t s);
/* hits whose positions may be resident on the device at once (awfm_gpu.hip: $AWFM_GPU_HIT_BUDGET_BYTES / 8, else a quarter
 * of the free device memory); `C` linkage like the rest of the shim */
extern "C" uint64_t awfmGpuHitBudget(const AwFmGpuIndex *g);
/* exclusive scan of the flags (counts[i] != 0) into dFlagOffsets[0..n] (awfm_gpu.hip); scratch as for the hit offsets */
enum AwFmReturnCode awfmGpuScanFlags(AwFmGpuIndex *g, const uint32_t *dCounts, uint64_t numQueries, uint64_t *dFlagOffsets,
                                     void *dScratch, hipStream_t s);
/* awfm_gpu_build.hip: level-wise construction of the deeper seed table into a new device buffer */
/* peakBytesOut (may be NULL): the most device memory the construction held at once (the table and the level below it) */
/* allocSecondsOut (may be NULL): wall seconds spent inside the hipMalloc calls of the levels -- on this pool a process's first
 * allocation of tens of GB sometimes takes seconds (memory the driver hands back from, or scrubs after, the process before) */
/* formatOut / bigOut (may be NULL: the table then has format 0 or 1): the format of the table's entries (DevIndex::deepNarrow)
 * and, for format 2, the 64-bit lengths of its long ranges ((bwtLength >> 11) + 2 words: DevIndex::deepBigBySp) */
bool awfmGpuBuildDeepSeedTable(const AwFmGpuIndex *g, unsigned deepK, void **tableOut, uint64_t *bytesOut, uint64_t *peakBytesOut = nullptr,
                               double *allocSecondsOut = nullptr, unsigned *formatOut = nullptr, void **bigOut = nullptr);
/* one table per k-mer length 1 .. maxDepth (<= 15), 8-byte entries {sp, length}, level d at entry awfmLengthTableAt(d) of one
 * allocation; entries in the deeper table's format, *bigOut: DevIndex::lengthBig of the wide one (awfm_gpu_build.hip) */
bool awfmGpuBuildLengthTables(const AwFmGpuIndex *g, unsigned maxDepth, void **tableOut, uint64_t *bytesOut, void **bigOut);
/* awfm_gpu_mixed.hip: the launches of awfm_mixed_lookup_kernel.h (a translation unit of their own) */
hipError_t awfmGpuLaunchMixedSample(const AwFmGpuIndex *g, hipStream_t s, const void *lengthTable, const uint8_t *dChars,
                                    const unsigned long long *off, unsigned long long nq, unsigned useNext, unsigned samples,
                                    unsigned long long *aliveOut, unsigned long long *verdictHost, unsigned searchNumber);
hipError_t awfmGpuLaunchMixedLookup(const AwFmGpuIndex *g, hipStream_t s, hipEvent_t start, hipEvent_t stop, const void *lengthTable,
                                    const uint8_t *dChars, const unsigned long long *off, unsigned long long nq, unsigned useNext,
                                    bool superInLds, const unsigned *sampleAlive, unsigned chooseOf, ulonglong2 *rng, unsigned *dCounts,
                                    unsigned *sparseCount, unsigned sparseCap, unsigned *sparseKmers, ulonglong2 *sparseRanges,
                                    unsigned long long *leftover, unsigned *leftoverCount, unsigned *kept);
hipError_t awfmGpuLaunchMixedTally(const AwFmGpuIndex *g, hipStream_t s, const void *lengthTable, const uint8_t *dChars,
                                   const unsigned long long *off, unsigned long long nq, unsigned useNext, unsigned long long *bits,
                                   unsigned long long lengthWords, unsigned long long deepWords, unsigned long long pairWords,
                                   unsigned long long nucWords);
unsigned awfmGpuMixedTouchLevels(void);
/* awfm_gpu_exact.hip: the launch of exactLookupSearchKernel (awfm_exact_lookup_kernel.h) */
hipError_t awfmGpuLaunchExactLookup(const AwFmGpuIndex *g, hipStream_t s, hipEvent_t start, hipEvent_t stop, const void *lengthTable,
                                    const uint8_t *dChars, const unsigned long long *off, unsigned fixedLength, unsigned long long nq,
                                    bool pairOff, ulonglong2 *rng, unsigned *dCounts, unsigned long long *leftover, unsigned *leftoverCount);
/* awfm_gpu_ordered.hip: awfmGpuSearch's exact ranges through the device-only tables: 1 = searched, 0 = does not apply (the
 * caller runs the general kernel), < 0 = -AwFmReturnCode */
int awfmGpuExactLookupSearch(AwFmGpuIndex *g, hipStream_t s, const uint8_t *dChars, const unsigned long long *off, uint32_t fixedLength,
                             unsigned long long nq, ulonglong2 *rng, uint32_t *dCounts);
/* awfm_gpu.hip: the suffix array (32-bit positions, `length` of them) the builder of this thread hands to the image it adopts
 * next (applyDenseSa takes it; whoever set it frees it when it is still there afterwards) */
extern thread_local void *awfmGpuDenseSaStash;
extern thread_local unsigned long long awfmGpuDenseSaStashLength;
extern thread_local bool awfmGpuDenseSaStashWide; /* 40-bit entries (DenseSa) instead of 32-bit ones */
/* awfm_gpu_ordered.hip: rewrites the 8-byte entries {sp, length} of a finished table as {sp, length16 | next16 << 16}
 * (DevIndex::deepNext) and returns the lengths that do not fit 16 bits in *bigOut ((bwtLength >> 15) + 1 words, indexed by
 * sp >> 15: DevIndex::deepBigBySp; *numBigOut: how many there are).  Needs the pair image.  1: done; 0: not applicable,
 * nothing was changed; -1: failed, the table is no longer usable. */
/* format 2 (awfmGpuBuildDeepSeedTable made the table and *big, its long lengths): only the entries' sixteen bits are
 * rewritten, *big is read and stays as it is, *numBigOut counts the long ranges */
int awfmGpuDeepSeedAddNext(AwFmGpuIndex *g, void *table, unsigned deepK, unsigned format, void **big, unsigned *numBigOut);
void awfmGpuSetError(const char *what);
void awfmGpuSetHipError(const char *what, hipError_t e);

#endif
