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
}  // namespace

extern "C" {

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
