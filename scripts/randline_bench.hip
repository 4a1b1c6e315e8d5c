// Random 128-B line gather ceiling on MI355X: every 8-lane group reads whole 128-B lines (16 B per lane)
// at hashed addresses of a table far larger than the Infinity Cache, UNROLL independent lines in flight
// per group.  Build: hipcc -O3 --offload-arch=gfx950 scripts/randline_bench.hip -o gpurun_out/randline
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
__device__ __forceinline__ u64 mix64(u64 z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
template <int UNROLL, int LINE_LANES>
__global__ void __launch_bounds__(256) gather(const uint4 *table, u64 lines, u64 readsPerGroup, unsigned *sink) {
  const u64 group = ((u64)blockIdx.x * 256 + threadIdx.x) / LINE_LANES;
  const unsigned g = threadIdx.x % LINE_LANES;
  unsigned acc = 0;
  for (u64 r = 0; r < readsPerGroup; r += UNROLL) {
    uint4 v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; k++) {
      const u64 line = mix64(group * 0x9E3779B97F4A7C15ull + r + k) % lines;
      v[k] = table[line * LINE_LANES + g];
    }
#pragma unroll
    for (int k = 0; k < UNROLL; k++) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
template <int UNROLL, int LINE_LANES>
void run(const uint4 *table, u64 bytes, unsigned *sink, int blocksPerCU) {
  const u64 lineBytes = 16ull * LINE_LANES, lines = bytes / lineBytes;
  const int grid = 256 * blocksPerCU;
  const u64 groups = (u64)grid * 256 / LINE_LANES;
  const u64 readsPerGroup = (800000000ull / groups) / UNROLL * UNROLL;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  gather<UNROLL, LINE_LANES><<<grid, 256>>>(table, lines, readsPerGroup / 8 + UNROLL, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  gather<UNROLL, LINE_LANES><<<grid, 256>>>(table, lines, readsPerGroup, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double n = (double)groups * readsPerGroup;
  printf("table %.2f GB  granule %3llu B  in-flight/group %d  blocks/CU %d : %.2f G granules/s  %.2f TB/s\n", bytes / 1e9,
         lineBytes, UNROLL, blocksPerCU, n / ms / 1e6, n * lineBytes / ms / 1e9);
}
int main(int argc, char **argv) {
  const u64 bytes = argc > 1 ? strtoull(argv[1], 0, 10) : 1600000000ull;
  uint4 *table;
  unsigned *sink;
  hipMalloc(&table, bytes);
  hipMalloc(&sink, 4);
  hipMemset(table, 1, bytes);
  for (int b : {4, 8}) {
    run<1, 8>(table, bytes, sink, b);
    run<2, 8>(table, bytes, sink, b);
    run<4, 8>(table, bytes, sink, b);
    run<8, 8>(table, bytes, sink, b);
  }
  run<4, 4>(table, bytes, sink, 8);   // 64-B granules
  run<4, 16>(table, bytes, sink, 8);  // 256-B granules
  run<2, 64>(table, bytes, sink, 8);  // 1-KiB granules (fully coalesced wave loads at random rows)
  return 0;
}
