/*
 * awfm_device.h -- device-side building blocks shared by awfm_gpu.hip (search /
 * locate) and awfm_gpu_build.hip (index construction): the kernel-argument view
 * of the device image, the 8-lane group rank/step primitives and the layout
 * conversion kernels.  C++/HIP only; the C ABI is include/awfm_gpu.h.
 */
#ifndef AWFM_DEVICE_H
#define AWFM_DEVICE_H

#include <hip/hip_runtime.h>

#include <mutex>
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

/* kernel-argument view of the device image */
struct DevIndex {
  const uint4 *blocks;
  const ulonglong2 *seed;
  const unsigned long long *sa; /* packed sampled SA viewed as 64-bit words */
  unsigned long long bwtLength;
  unsigned long long sentinelPos; /* BWT position holding '$' */
  unsigned long long seedLen;
  const unsigned long long *prefixSums; /* 24 words in device memory */
  unsigned int saRatio;
  unsigned int saShift; /* log2(saRatio) when it is a power of two, else 0xFFFFFFFF */
  unsigned int saWidth;
  unsigned int seedK;
  /* optional device-only deeper seed table (nucleotide): entry of a deepK-mer = the range the reference
   * algorithm reaches after the seed lookup and (deepK - seedK) extension steps that stop at the first
   * invalid range; NULL when not built */
  const ulonglong2 *deepSeed;
  unsigned int deepK;
};

/* a nucleotide query prepared for the ordered search path (awfm_ordered_kernel.h) */
struct QueryRec {
  unsigned long long codes; /* 2-bit letter codes of the k-mer, last character in bits 1..0 */
  unsigned int index;       /* query number in the batch */
  unsigned int length;      /* characters (1..32), or 0xFFFFFFFF: left to the general kernel */
};

constexpr int kThreads = 256;
constexpr int kGroupsPerBlock = kThreads / 8;

/* ------------------------------------------------------------------ device helpers */

template <int CTRL>
__device__ __forceinline__ unsigned dppMove(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

/* sum over the 8 lanes of a group; every lane gets the total */
__device__ __forceinline__ unsigned groupSum8(unsigned v) {
  v += dppMove<0xB1>(v);  /* quad_perm [1,0,3,2] */
  v += dppMove<0x4E>(v);  /* quad_perm [2,3,0,1] */
  v += dppMove<0x141>(v); /* row_half_mirror: lane i <- lane 7-i of its half row */
  return v;
}

__device__ __forceinline__ unsigned long long groupSum8u64(unsigned long long v) {
#pragma unroll
  for (int stage = 0; stage < 3; stage++) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32), olo, ohi;
    if (stage == 0) {
      olo = dppMove<0xB1>(lo);
      ohi = dppMove<0xB1>(hi);
    } else if (stage == 1) {
      olo = dppMove<0x4E>(lo);
      ohi = dppMove<0x4E>(hi);
    } else {
      olo = dppMove<0x141>(lo);
      ohi = dppMove<0x141>(hi);
    }
    v += ((unsigned long long)ohi << 32) | olo;
  }
  return v;
}

/* bits 0..(p - 32*piece) of a 32-position slice, clamped: the slice's share of
 * the inclusive prefix mask of ref src/AwFmSimdConfig.c:89-114 */
__device__ __forceinline__ unsigned sliceMask(unsigned p, unsigned piece) {
  int bits = (int)p - (int)(piece * 32) + 1;
  bits = bits < 0 ? 0 : (bits > 32 ? 32 : bits);
  return (unsigned)((1ull << bits) - 1ull);
}

/* ---- nucleotide ---- */

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

__device__ __forceinline__ unsigned nucOccSlice(const uint4 &pc, const PlaneSel3 &s) {
  return ((pc.x ^ s.x0) | s.d0) & ((pc.y ^ s.x1) | s.d1) & ((pc.z ^ s.x2) | s.d2);
}

/* base count of `letter` before block `blk` from the count words spread over the group */
__device__ __forceinline__ unsigned long long nucBase(const uint4 &pc, unsigned letter, unsigned long long blk,
                                                      unsigned long long sentinelPos, unsigned g) {
  if (letter < 4u) {
    const unsigned lo = (unsigned)__shfl((int)pc.w, (int)(2u * letter), 8);
    const unsigned hi = (unsigned)__shfl((int)pc.w, (int)(2u * letter + 1u), 8);
    return ((unsigned long long)hi << 32) | lo;
  }
  /* X (or anything else): everything before the block that is not A,C,G,T,$ */
  const unsigned long long part = (g & 1u) ? ((unsigned long long)pc.w << 32) : (unsigned long long)pc.w;
  const unsigned long long acgt = groupSum8u64(part);
  const unsigned long long before = blk * 256ull;
  return before - acgt - (sentinelPos < before ? 1ull : 0ull);
}

/* rank arithmetic of one backward step once the two pieces are in registers
 * (ref src/AwFmSearch.c:42-103); pc1 == pc0 when sp-1 and ep share a block */
__device__ __forceinline__ void nucStepFromPieces(const DevIndex &ix, const unsigned long long *sC, unsigned letter,
                                                  const uint4 &pc0, const uint4 &pc1, unsigned long long &sp,
                                                  unsigned long long &ep, unsigned g) {
  const unsigned long long q0 = sp - 1ull, q1 = ep;
  const PlaneSel3 sel = nucPlaneSel(letter);
  const unsigned n0 = __popc(nucOccSlice(pc0, sel) & sliceMask((unsigned)q0 & 255u, g));
  const unsigned n1 = __popc(nucOccSlice(pc1, sel) & sliceMask((unsigned)q1 & 255u, g));
  const unsigned packed = groupSum8(n0 | (n1 << 16));
  const unsigned long long base0 = nucBase(pc0, letter, q0 >> 8, ix.sentinelPos, g);
  const unsigned long long base1 = nucBase(pc1, letter, q1 >> 8, ix.sentinelPos, g);
  const unsigned long long c = sC[letter];
  sp = c + base0 + (packed & 0xFFFFu);
  ep = c + base1 + (packed >> 16) - 1ull;
}

/* one backward step for the group's query: loads + rank */
__device__ __forceinline__ void nucStep(const DevIndex &ix, const unsigned long long *sC, unsigned letter,
                                        unsigned long long &sp, unsigned long long &ep, unsigned g) {
  const unsigned long long blk0 = (sp - 1ull) >> 8, blk1 = ep >> 8;
  /* both loads are issued before either is consumed (copying pc0 into pc1 first would serialise them) */
  const uint4 pc0 = ix.blocks[blk0 * 8ull + g];
  uint4 other = make_uint4(0u, 0u, 0u, 0u);
  if (blk1 != blk0) other = ix.blocks[blk1 * 8ull + g];
  const uint4 pc1 = blk1 != blk0 ? other : pc0;
  nucStepFromPieces(ix, sC, letter, pc0, pc1, sp, ep, g);
}

/* ---- amino ---- */

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

__device__ __forceinline__ unsigned aminoLetterIndex(const AminoShared &t, unsigned c) {
  return c == '$' ? 21u : (unsigned)t.letterOfAscii[c & 31u];
}
__device__ __forceinline__ bool aminoIsAmbiguous(unsigned c) {
  const unsigned l = c | 0x20u; /* equals tolower(c) whenever the result is z, x or b */
  return ((l == 'z') | (l == 'x') | (l == 'b')) != 0;
}

/* An amino piece is two 16-B loads kept as plain uint4 values: lo = {b0,b1,b2,b3},
 * hi = {b4,c0,c1,c2} (plane words of this lane's 32 positions, then the base
 * counts of letters 3k, 3k+1, 3k+2). */
__device__ __forceinline__ unsigned aminoLiteral(unsigned plane, unsigned ones, unsigned zeros, unsigned j) {
  const unsigned x = 0u - ((zeros >> j) & 1u);
  const unsigned d = (((ones | zeros) >> j) & 1u) - 1u;
  return (plane ^ x) | d;
}

__device__ __forceinline__ unsigned aminoOccSlice(const uint4 &lo, const uint4 &hi, unsigned ones, unsigned zeros) {
  return aminoLiteral(lo.x, ones, zeros, 0) & aminoLiteral(lo.y, ones, zeros, 1) &
         aminoLiteral(lo.z, ones, zeros, 2) & aminoLiteral(lo.w, ones, zeros, 3) &
         aminoLiteral(hi.x, ones, zeros, 4);
}

__device__ __forceinline__ unsigned long long aminoBase(const uint4 &hi, unsigned letter) {
  /* pick count word 1+letter%3 of `hi` with shifts (a select chain on vector
   * components makes hipcc spill the vector to LDS for dynamic indexing) */
  const unsigned slot = letter % 3u;
  const unsigned long long c01 = ((unsigned long long)hi.z << 32) | hi.y;
  const unsigned long long c2x = hi.w;
  const unsigned mine = (unsigned)((slot == 2u ? c2x : c01) >> (slot == 1u ? 32u : 0u));
  return (unsigned)__shfl((int)mine, (int)(letter / 3u), 8);
}

/* rank arithmetic of one amino backward step from loaded pieces (ref src/AwFmSearch.c:105-159) */
__device__ __forceinline__ void aminoStepFromPieces(const unsigned long long *sC, const AminoShared &t, unsigned letter,
                                                    const uint4 &lo0, const uint4 &hi0, const uint4 &lo1,
                                                    const uint4 &hi1, unsigned long long &sp, unsigned long long &ep,
                                                    unsigned g) {
  const unsigned long long q0 = sp - 1ull, q1 = ep;
  const unsigned pm = t.planeMask[letter < 24u ? letter : 23u];
  const unsigned ones = pm & 0xFFu, zeros = pm >> 8;
  const unsigned n0 = __popc(aminoOccSlice(lo0, hi0, ones, zeros) & sliceMask((unsigned)q0 & 255u, g));
  const unsigned n1 = __popc(aminoOccSlice(lo1, hi1, ones, zeros) & sliceMask((unsigned)q1 & 255u, g));
  const unsigned packed = groupSum8(n0 | (n1 << 16));
  const unsigned long long c = sC[letter];
  sp = c + aminoBase(hi0, letter) + (packed & 0xFFFFu);
  ep = c + aminoBase(hi1, letter) + (packed >> 16) - 1ull;
}

__device__ __forceinline__ void aminoStep(const DevIndex &ix, const unsigned long long *sC, const AminoShared &t,
                                          unsigned letter, unsigned long long &sp, unsigned long long &ep,
                                          unsigned g) {
  const unsigned long long blk0 = (sp - 1ull) >> 8, blk1 = ep >> 8;
  const uint4 lo0 = ix.blocks[blk0 * 16ull + 2u * g];
  const uint4 hi0 = ix.blocks[blk0 * 16ull + 2u * g + 1u];
  uint4 otherLo = make_uint4(0u, 0u, 0u, 0u), otherHi = otherLo;
  if (blk1 != blk0) {
    otherLo = ix.blocks[blk1 * 16ull + 2u * g];
    otherHi = ix.blocks[blk1 * 16ull + 2u * g + 1u];
  }
  const uint4 lo1 = blk1 != blk0 ? otherLo : lo0, hi1 = blk1 != blk0 ? otherHi : hi0;
  aminoStepFromPieces(sC, t, letter, lo0, hi0, lo1, hi1, sp, ep, g);
}


/* ------------------------------------------------------------------ image build kernels */

/* reference-layout blocks -> device layout; also finds the sentinel's BWT position */
__global__ void relayoutNucKernel(const unsigned long long *__restrict__ ref, unsigned long long numBlocks,
                                  unsigned long long bwtLength, uint4 *__restrict__ out,
                                  unsigned long long *__restrict__ sentinelPos) {
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long blk = t >> 3;
  const unsigned k = (unsigned)t & 7u;
  if (blk >= numBlocks) return;
  const unsigned long long *src = ref + blk * 20ull; /* 160 B = 20 words: planes [3][4], counts [8] */
  const unsigned half = (k & 1u) * 32u;
  const unsigned b0 = (unsigned)(src[0 + (k >> 1)] >> half);
  const unsigned b1 = (unsigned)(src[4 + (k >> 1)] >> half);
  const unsigned b2 = (unsigned)(src[8 + (k >> 1)] >> half);
  const unsigned cw = (unsigned)(src[12 + (k >> 1)] >> half);
  out[blk * 8ull + k] = make_uint4(b0, b1, b2, cw);
  unsigned sentinelBits = b2 & ~b1 & ~b0; /* code 100b, ref src/AwFmLetter.c:44-47 */
  if (sentinelBits) {
    const unsigned long long pos = blk * 256ull + k * 32u + (unsigned)(__ffs((int)sentinelBits) - 1);
    if (pos < bwtLength) *sentinelPos = pos;
  }
}

__global__ void relayoutAminoKernel(const unsigned long long *__restrict__ ref, unsigned long long numBlocks,
                                    unsigned long long bwtLength, uint4 *__restrict__ out,
                                    unsigned long long *__restrict__ sentinelPos) {
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long blk = t >> 3;
  const unsigned k = (unsigned)t & 7u;
  if (blk >= numBlocks) return;
  const unsigned long long *src = ref + blk * 44ull; /* 352 B = 44 words: planes [5][4], counts [24] */
  const unsigned half = (k & 1u) * 32u;
  unsigned b[5];
  for (int j = 0; j < 5; j++) b[j] = (unsigned)(src[4 * j + (k >> 1)] >> half);
  unsigned c[3];
  for (unsigned s = 0; s < 3; s++) {
    const unsigned letter = 3u * k + s;
    c[s] = letter < 21u ? (unsigned)src[20 + letter] : 0u;
  }
  out[blk * 16ull + 2u * k] = make_uint4(b[0], b[1], b[2], b[3]);
  out[blk * 16ull + 2u * k + 1u] = make_uint4(b[4], c[0], c[1], c[2]);
  unsigned sentinelBits = ~(b[0] | b[1] | b[2] | b[3] | b[4]); /* code 00000 */
  while (sentinelBits) {
    const unsigned bit = (unsigned)(__ffs((int)sentinelBits) - 1);
    sentinelBits &= sentinelBits - 1u;
    const unsigned long long pos = blk * 256ull + k * 32u + bit;
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
  void *dSeed = nullptr;
  void *dSa = nullptr;
  void *dPrefix = nullptr;
  void *dDeepSeed = nullptr;
  uint64_t deepSeedBytes = 0;
  void *dDenseSa = nullptr; /* optional full suffix array, 32-bit entries */
  uint64_t denseSaBytes = 0;
  uint64_t deviceBytes = 0;
  uint64_t numBlocks = 0;
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
  /* ordered search path (awfm_gpu_ordered.hip): grow-only scratch shared by all searches on this image;
   * the event orders its re-use across streams */
  int orderMode = -1; /* -1 auto, 0 off, 1 on */
  std::mutex orderMutex;
  void *dOrder = nullptr;
  size_t orderBytes = 0;
  hipEvent_t orderEvent = nullptr;
  bool orderEventRecorded = false;
  hipEvent_t orderTiming[2] = {nullptr, nullptr}; /* around orderedSearchKernel when $AWFM_GPU_TIME_ORDERED is set */
  bool orderTimed = false;
  std::mutex aosMutex;       /* serialises the AoS entry points (they share the pinned buffers) */
  void *pinned[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t pinnedBytes[4] = {0, 0, 0, 0};
};

/* 32-bit BWT positions in the kernels: exact whenever bwtLength < 2^32 (ref src/AwFmIndex.h:88-91 is 64-bit
 * throughout; the NARROW = false instantiations are that arithmetic) */
inline bool awfmImageNarrow(const AwFmGpuIndex *g) { return !g->forceWide && g->dev.bwtLength < (1ull << 32); }

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

/* ordered hits-only search; 1 = searched, 0 = does not apply, <0 = -AwFmReturnCode (awfm_gpu_ordered.hip) */
int awfmGpuOrderedSearch(AwFmGpuIndex *g, hipStream_t s, const uint8_t *dChars, const unsigned long long *off,
                         uint32_t fixedLength, unsigned long long nq, ulonglong2 *rng, uint32_t *dCounts);

/* adopts device buffers that already hold a complete image (used by the GPU builder) */
AwFmGpuIndex *awfmGpuIndexAdopt(const struct AwFmIndex *index, int device, void *dBlocks, void *dSeed, void *dSa,
                                void *dPrefix, unsigned long long sentinelPos, uint64_t deviceBytes);
void awfmGpuIndexRegister(const struct AwFmIndex *index, AwFmGpuIndex *g);
/* awfm_gpu_build.hip: level-wise construction of the deeper seed table into a new device buffer */
bool awfmGpuBuildDeepSeedTable(const AwFmGpuIndex *g, unsigned deepK, void **tableOut, uint64_t *bytesOut);
void awfmGpuSetError(const char *what);
void awfmGpuSetHipError(const char *what, hipError_t e);

#endif
