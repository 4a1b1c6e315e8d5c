// Cost of ordering 100 M query records by a 16-bit (or narrower) key with rocPRIM radix_sort_pairs.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/sort_probe.hip -o gpurun_out/sort_probe
// MI355X: u16 key + 16-B record 2.14 ms (16 bits), 1.98 ms (12), 1.20 ms (8); u16 key + 4-B index 1.42 / 1.40 / 0.77 ms.
#include <cstring>
#include <cstdio>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdlib>
typedef unsigned long long u64;
struct Rec { u64 codes; unsigned index; unsigned len; };
__global__ void fill(unsigned short *k16, unsigned *i32, Rec *v, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 z = (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  k16[i] = (unsigned short)z;
  i32[i] = (unsigned)i;
  v[i].codes = z; v[i].index = (unsigned)i; v[i].len = 21;
}
template <class K, class V>
float timeSort(K *kin, K *kout, V *vin, V *vout, u64 n, unsigned bits) {
  size_t tb = 0;
  rocprim::radix_sort_pairs(nullptr, tb, kin, kout, vin, vout, n, 0, bits);
  void *tmp; hipMalloc(&tmp, tb);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9;
  for (int r = 0; r < 4; r++) {
    hipEventRecord(a);
    rocprim::radix_sort_pairs(tmp, tb, kin, kout, vin, vout, n, 0, bits);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (r && ms < best) best = ms;
  }
  hipFree(tmp);
  return best;
}
int main() {
  const u64 n = 100000000ull;
  unsigned short *k16, *k16o; unsigned *i32, *i32o; Rec *v, *vo;
  hipMalloc(&k16, n * 2); hipMalloc(&k16o, n * 2);
  hipMalloc(&i32, n * 4); hipMalloc(&i32o, n * 4);
  hipMalloc(&v, n * sizeof(Rec)); hipMalloc(&vo, n * sizeof(Rec));
  fill<<<(n + 255) / 256, 256>>>(k16, i32, v, n);
  hipDeviceSynchronize();
  for (unsigned bits : {16u, 12u, 8u}) {
    printf("u16 key, 16-B value, %2u bits: %.2f ms\n", bits, timeSort(k16, k16o, v, vo, n, bits));
    printf("u16 key,  4-B value, %2u bits: %.2f ms\n", bits, timeSort(k16, k16o, i32, i32o, n, bits));
  }
  return 0;
}
