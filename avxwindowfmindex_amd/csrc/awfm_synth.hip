/*
 * awfm_synth.hip -- seeded synthetic texts and query sets generated on the
 * device.  Bit-identical to avxwindowfmindex_amd/synth.py:
 * splitmix64 is counter based, so character i of a text and character c of
 * query j are pure functions of (seed, i) / (seed_q, j, c).  Used by bench.py
 * and the full-size tests; not part of the reference's API.
 */
#include <hip/hip_runtime.h>

#include "awfm_device.h"

namespace {
typedef unsigned long long u64;
constexpr u64 kGolden = 0x9E3779B97F4A7C15ull;

__device__ __forceinline__ u64 mix64(u64 z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__constant__ unsigned char kDna[4] = {'a', 'c', 'g', 't'};
__constant__ unsigned char kAmino[20] = {'a', 'c', 'd', 'e', 'f', 'g', 'h', 'i', 'k', 'l',
                                         'm', 'n', 'p', 'q', 'r', 's', 't', 'v', 'w', 'y'};

__device__ __forceinline__ unsigned char letterOf(u64 z, int amino) { return amino ? kAmino[z % 20ull] : kDna[z % 4ull]; }

__global__ void synthTextKernel(unsigned char *out, u64 start, u64 count, u64 seed, int amino) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride)
    out[i] = letterOf(mix64(seed + (start + i + 1ull) * kGolden), amino);
}

__global__ void synthRandomQueriesKernel(unsigned char *out, u64 first, u64 count, unsigned length, u64 seedQ, int amino) {
  const u64 total = count * length, stride = (u64)gridDim.x * blockDim.x;
  for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const u64 j = t / length, c = t % length;
    const u64 q = mix64(seedQ + first + j);
    out[t] = letterOf(mix64(q + (c + 1ull) * kGolden), amino);
  }
}

__global__ void synthPlantedQueriesKernel(unsigned char *out, u64 first, u64 count, unsigned length, u64 seedQ,
                                          const unsigned char *text, u64 n) {
  const u64 total = count * length, stride = (u64)gridDim.x * blockDim.x;
  for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const u64 j = t / length, c = t % length;
    const u64 q = mix64(seedQ + first + j);
    const u64 offset = mix64(q + kGolden) % (n - length + 1ull);
    out[t] = text[offset + c];
  }
}
/* mixed-length set: L_j = lo + mix(q_j + 2^63 + G) % (hi-lo+1); even ids random, odd ids planted */
__global__ void synthMixedLengthsKernel(u64 *lengths, u64 first, u64 count, unsigned lo, unsigned hi, u64 seedQ) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += stride) {
    const u64 q = mix64(seedQ + first + j);
    lengths[j] = lo + mix64(q + (1ull << 63) + kGolden) % (u64)(hi - lo + 1u);
  }
}

__global__ void synthMixedQueriesKernel(unsigned char *out, const u64 *offsets, u64 first, u64 count, u64 seedQ,
                                        const unsigned char *text, u64 n, int amino) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += stride) {
    const u64 q = mix64(seedQ + first + j);
    const u64 o = offsets[j], L = offsets[j + 1] - o;
    if (((first + j) & 1ull) == 0ull) {
      for (u64 c = 0; c < L; c++) out[o + c] = letterOf(mix64(q + (c + 1ull) * kGolden), amino);
    } else {
      const u64 s = mix64(q + kGolden) % (n - L + 1ull);
      for (u64 c = 0; c < L; c++) out[o + c] = text[s + c];
    }
  }
}

/*
 * Genome-shaped text (avxwindowfmindex_amd/synth.py: genome_text): what a uniform text lacks and an assembly like GRCh38
 * has -- interspersed repeat families, tandem repeats, long runs of N.  Character i is a pure function of (seed, i, n):
 * the text is cut into blocks of 1024 characters, and a block is, by its hash,
 *   10 %  a window into the endless repetition of a 300-character consensus (a short interspersed family: about
 *         n / 3000 copies), every character replaced by a random one with probability 10 %,
 *   15 %  a window of a 6000-character consensus at a random offset (a long family, cut copies), 5 % divergence,
 *    3 %  a tandem repeat of a unit of 2..64 characters of its own, 2 % divergence,
 *   72 %  unique sequence (the characters of the uniform text of the same seed);
 * on top, 24 runs of 'n' with seeded starts and lengths between n/6400 and n/64 (at most 10^5 .. 10^7 characters).
 */
constexpr unsigned kGenomeBlock = 1024, kGenomeRuns = 24;
constexpr u64 kSaltBlock = 0xB10C5A17ull, kSaltFamA = 0xFA111A5ull, kSaltFamB = 0xFA111B5ull, kSaltDiv = 0xD17E26E5ull,
              kSaltRunStart = 0x52554E53ull, kSaltRunLen = 0x52554E4Cull;

__device__ __forceinline__ unsigned char genomeChar(u64 i, u64 n, u64 seed, const u64 *runStart, const u64 *runLen) {
  for (unsigned r = 0; r < kGenomeRuns; r++)
    if (i - runStart[r] < runLen[r]) return 'n';
  const u64 b = i / kGenomeBlock, j = i % kGenomeBlock;
  const u64 hb = mix64(seed + kSaltBlock + (b + 1ull) * kGolden);
  const unsigned kind = (unsigned)(hb % 100ull);
  const u64 pick = hb >> 8;
  unsigned char base;
  unsigned permille;
  if (kind < 10u) {
    base = kDna[mix64(seed + kSaltFamA + ((pick + j) % 300ull + 1ull) * kGolden) % 4ull];
    permille = 100u;
  } else if (kind < 25u) {
    base = kDna[mix64(seed + kSaltFamB + ((pick + j) % 6000ull + 1ull) * kGolden) % 4ull];
    permille = 50u;
  } else if (kind < 28u) {
    const u64 unit = 2ull + pick % 63ull;
    base = kDna[mix64(hb + (j % unit + 1ull) * kGolden) % 4ull];
    permille = 20u;
  } else {
    return kDna[mix64(seed + (i + 1ull) * kGolden) % 4ull];
  }
  const u64 r = mix64(seed + kSaltDiv + (i + 1ull) * kGolden);
  return (unsigned)(r % 1000ull) < permille ? kDna[(r >> 16) % 4ull] : base;
}

__global__ void synthGenomeTextKernel(unsigned char *out, u64 n, u64 seed) {
  __shared__ u64 runStart[kGenomeRuns], runLen[kGenomeRuns];
  if (threadIdx.x < kGenomeRuns) {
    u64 longest = n / 64ull < 10000000ull ? n / 64ull : 10000000ull;
    if (longest < 1ull) longest = 1ull;
    u64 shortest = longest / 100ull < 100000ull ? longest / 100ull : 100000ull;
    if (shortest < 1ull) shortest = 1ull;
    runStart[threadIdx.x] = mix64(seed + kSaltRunStart + threadIdx.x) % n;
    runLen[threadIdx.x] = shortest + mix64(seed + kSaltRunLen + threadIdx.x) % (longest - shortest + 1ull);
  }
  __syncthreads();
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = genomeChar(i, n, seed, runStart, runLen);
}

/* k-mers copied from the text like synthPlantedQueriesKernel, but a character that is not a,c,g,t (the 'n' of a
 * genome-shaped text) is replaced by a seeded random letter: a read has no 21 N in a row, and a k-mer of N would match
 * every window of every N run */
__global__ void synthPlantedCleanKernel(unsigned char *out, u64 first, u64 count, unsigned length, u64 seedQ,
                                        const unsigned char *text, u64 n) {
  const u64 total = count * length, stride = (u64)gridDim.x * blockDim.x;
  for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const u64 j = t / length, c = t % length;
    const u64 q = mix64(seedQ + first + j);
    const u64 offset = mix64(q + kGolden) % (n - length + 1ull);
    const unsigned char ch = text[offset + c];
    const bool plain = ch == 'a' || ch == 'c' || ch == 'g' || ch == 't';
    out[t] = plain ? ch : kDna[mix64(q + (c + 2ull) * kGolden) % 4ull];
  }
}
/* k-mers drawn from the UNIQUE sequence of a genome-shaped text (the 72 % of its 1024-character blocks that are no repeat
 * family's and no tandem repeat's, outside the runs of 'n'): offset number t of k-mer j is mix64(q + (t + 1) kGolden) mod
 * (n - length + 1), and the k-mer takes the first of up to 64 whose window lies in unique blocks and holds only a,c,g,t -- so
 * that every k-mer has a hit at a known offset and few besides (a 21-mer out of a repeat family has 10^5): the batch a
 * locate of k-mers "drawn from the text" can be timed on.  One thread per k-mer. */
__global__ void synthPlantedUniqueKernel(unsigned char *out, u64 first, u64 count, unsigned length, u64 seedQ,
                                         const unsigned char *text, u64 n, u64 textSeed, u64 *offsetsOut) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += stride) {
    const u64 q = mix64(seedQ + first + j);
    u64 offset = 0;
    for (u64 t = 0; t < 64ull; t++) {
      offset = mix64(q + (t + 1ull) * kGolden) % (n - length + 1ull);
      const u64 b0 = offset / kGenomeBlock, b1 = (offset + length - 1ull) / kGenomeBlock;
      bool ok = mix64(textSeed + kSaltBlock + (b0 + 1ull) * kGolden) % 100ull >= 28ull &&
                mix64(textSeed + kSaltBlock + (b1 + 1ull) * kGolden) % 100ull >= 28ull;
      for (unsigned c = 0; ok && c < length; c++) {
        const unsigned char ch = text[offset + c];
        ok = ch == 'a' || ch == 'c' || ch == 'g' || ch == 't';
      }
      if (ok) break;
    }
    for (unsigned c = 0; c < length; c++) out[j * length + c] = text[offset + c];
    if (offsetsOut) offsetsOut[j] = offset;
  }
}
}  // namespace

extern "C" {

enum AwFmReturnCode awfmGpuSynthPlantedQueriesUnique(uint8_t *dOut, uint64_t first, uint64_t count, uint32_t length, uint64_t seedQ,
                                                     const uint8_t *dText, uint64_t textLength, uint64_t textSeed,
                                                     uint64_t *dOffsetsOut, void *stream) {
  if (!dOut || !dText) return AwFmNullPtrError;
  if (count == 0 || length == 0) return AwFmSuccess;
  if (textLength < length) return AwFmIllegalPositionError;
  hipLaunchKernelGGL(synthPlantedUniqueKernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, dOut, (u64)first, (u64)count, length,
                     (u64)seedQ, dText, (u64)textLength, (u64)textSeed, (u64 *)dOffsetsOut);
  return hipGetLastError() == hipSuccess ? AwFmSuccess : AwFmGeneralFailure;
}

enum AwFmReturnCode awfmGpuSynthGenomeText(uint8_t *dOut, uint64_t length, uint64_t seed, void *stream) {
  if (!dOut) return AwFmNullPtrError;
  if (length == 0) return AwFmSuccess;
  hipLaunchKernelGGL(synthGenomeTextKernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, dOut, (u64)length, (u64)seed);
  return hipGetLastError() == hipSuccess ? AwFmSuccess : AwFmGeneralFailure;
}

enum AwFmReturnCode awfmGpuSynthPlantedQueriesClean(uint8_t *dOut, uint64_t first, uint64_t count, uint32_t length,
                                                    uint64_t seedQ, const uint8_t *dText, uint64_t textLength, void *stream) {
  if (!dOut || !dText) return AwFmNullPtrError;
  if (count == 0 || length == 0) return AwFmSuccess;
  if (textLength < length) return AwFmIllegalPositionError;
  hipLaunchKernelGGL(synthPlantedCleanKernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, dOut, (u64)first, (u64)count,
                     length, (u64)seedQ, dText, (u64)textLength);
  return hipGetLastError() == hipSuccess ? AwFmSuccess : AwFmGeneralFailure;
}

enum AwFmReturnCode awfmGpuSynthMixedLengths(uint64_t *dLengths, uint64_t first, uint64_t count, uint32_t lo,
                                             uint32_t hi, uint64_t seedQ, void *stream) {
  if (!dLengths || hi < lo) return AwFmNullPtrError;
  if (count == 0) return AwFmSuccess;
  hipLaunchKernelGGL(synthMixedLengthsKernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, (u64 *)dLengths, (u64)first,
                     (u64)count, lo, hi, (u64)seedQ);
  return hipGetLastError() == hipSuccess ? AwFmSuccess : AwFmGeneralFailure;
}

enum AwFmReturnCode awfmGpuSynthMixedQueries(uint8_t *dOut, const uint64_t *dOffsets, uint64_t first, uint64_t count,
                                             uint64_t seedQ, const uint8_t *dText, uint64_t textLength, int amino,
                                             void *stream) {
  if (!dOut || !dOffsets || !dText) return AwFmNullPtrError;
  if (count == 0) return AwFmSuccess;
  hipLaunchKernelGGL(synthMixedQueriesKernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, dOut, (const u64 *)dOffsets,
                     (u64)first, (u64)count, (u64)seedQ, dText, (u64)textLength, amino);
  return hipGetLastError() == hipSuccess ? AwFmSuccess : AwFmGeneralFailure;
}

enum AwFmReturnCode awfmGpuSynthText(uint8_t *dOut, uint64_t start, uint64_t count, uint64_t seed, int amino,
                                     void *stream) {
  if (!dOut) return AwFmNullPtrError;
  if (count == 0) return AwFmSuccess;
  hipLaunchKernelGGL(synthTextKernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, dOut, (u64)start, (u64)count,
                     (u64)seed, amino);
  if (hipGetLastError() != hipSuccess) return AwFmGeneralFailure;
  return AwFmSuccess;
}

enum AwFmReturnCode awfmGpuSynthRandomQueries(uint8_t *dOut, uint64_t first, uint64_t count, uint32_t length,
                                              uint64_t seedQ, int amino, void *stream) {
  if (!dOut) return AwFmNullPtrError;
  if (count == 0 || length == 0) return AwFmSuccess;
  hipLaunchKernelGGL(synthRandomQueriesKernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, dOut, (u64)first,
                     (u64)count, length, (u64)seedQ, amino);
  if (hipGetLastError() != hipSuccess) return AwFmGeneralFailure;
  return AwFmSuccess;
}

enum AwFmReturnCode awfmGpuSynthPlantedQueries(uint8_t *dOut, uint64_t first, uint64_t count, uint32_t length,
                                               uint64_t seedQ, const uint8_t *dText, uint64_t textLength, void *stream) {
  if (!dOut || !dText) return AwFmNullPtrError;
  if (count == 0 || length == 0) return AwFmSuccess;
  if (textLength < length) return AwFmIllegalPositionError;
  hipLaunchKernelGGL(synthPlantedQueriesKernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, dOut, (u64)first,
                     (u64)count, length, (u64)seedQ, dText, (u64)textLength);
  if (hipGetLastError() != hipSuccess) return AwFmGeneralFailure;
  return AwFmSuccess;
}

}  // extern "C"
