// Issue cost of the integer VALU instructions the search kernels are made of, wave64 on gfx950: cycles per instruction
// per SIMD with 1, 2, 4 and 8 waves resident per SIMD (independent instructions, four chains per lane).
// MI355X (cycles at the nominal 2.4 GHz, 8 waves per SIMD): v_and_b32 2.4, v_add_u32 2.6, v_lshrrev_b32 2.3, v_bitop3_b32
// 2.8, v_fma_f32 2.5 -- full rate, 2 cycles at the clock the chip holds; v_bcnt_u32_b32 4.1, v_mov_b32_dpp 4.2,
// v_add_u32_dpp 4.1, v_cndmask_b32 (SGPR mask) 4.2, v_alignbit_b32 4.2, v_lshrrev_b64 4.3 -- half rate; one wave alone
// pays 5.0-5.5 for any of them.  (v_cndmask_b32 reading vcc from inline asm: 23 -- an artefact of the probe, vcc is never
// written.)
// Build: hipcc -O3 --offload-arch=gfx950 scripts/valu_probe.hip -o scripts/valu_probe.bin
#include <cstdio>
#include <hip/hip_runtime.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int OP>
__global__ void __launch_bounds__(256) probe(unsigned *out, int iters, unsigned seed) {
  unsigned a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 77u, d = b + 1234567u;
  const unsigned m = seed | 1u, n = ~seed;
  const unsigned long long mask = 0x5555555555555555ull * (seed | 1u);
  unsigned long long wide0 = a * 0x100000001ull, wide1 = b * 0x100000001ull;
  for (int i = 0; i < iters; i++) {
    if (OP == 0) { REP64(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
    if (OP == 1) { REP64(asm volatile("v_bcnt_u32_b32 %0, %0, %4\n v_bcnt_u32_b32 %1, %1, %4\n v_bcnt_u32_b32 %2, %2, %4\n v_bcnt_u32_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
    if (OP == 2) { REP64(asm volatile("v_bitop3_b32 %0, %0, %4, %5 bitop3:0xE4\n v_bitop3_b32 %1, %1, %4, %5 bitop3:0xE4\n v_bitop3_b32 %2, %2, %4, %5 bitop3:0xE4\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0xE4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(n));) }
    if (OP == 3) { REP64(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
    if (OP == 4) { REP64(asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (OP == 5) { REP64(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (OP == 6) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(n));) }
    if (OP == 7) { REP64(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc");) }
    if (OP == 8) { REP64(asm volatile("v_cndmask_b32_e64 %0, %0, %4, %5\n v_cndmask_b32_e64 %1, %1, %4, %5\n v_cndmask_b32_e64 %2, %2, %4, %5\n v_cndmask_b32_e64 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "s"(mask));) }
    if (OP == 9) { REP64(asm volatile("v_alignbit_b32 %0, %0, %4, 7\n v_alignbit_b32 %1, %1, %4, 7\n v_alignbit_b32 %2, %2, %4, 7\n v_alignbit_b32 %3, %3, %4, 7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
    if (OP == 10) { REP64(asm volatile("v_lshrrev_b64 %0, 3, %0\n v_lshrrev_b64 %1, 3, %1" : "+v"(wide0), "+v"(wide1));) }
    if (OP == 11) { REP64(asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ (unsigned)wide0 ^ (unsigned)wide1;
}

template <int OP>
void run(const char *name, unsigned *out, int numCUs, double perAsm = 4.0) {
  const int iters = 2000; // 2000 * 64 * 4 = 512000 instructions per wave
  for (int wavesPerSimd : {1, 2, 4, 8}) {
    const int blocks = numCUs * wavesPerSimd; // 256 threads = 4 waves = one per SIMD of a CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<OP><<<blocks, 256>>>(out, 10, 1u);
    hipEventRecord(e0);
    probe<OP><<<blocks, 256>>>(out, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instrPerSimd = (double)iters * 64.0 * perAsm * wavesPerSimd;
    printf("%-18s %d waves/SIMD: %.3f ms  -> %.2f cycles per wave64 instruction per SIMD at 2.4 GHz\n", name, wavesPerSimd, ms,
           ms * 1e-3 * 2.4e9 / instrPerSimd);
  }
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  unsigned *out; hipMalloc(&out, (size_t)p.multiProcessorCount * 8 * 256 * 4);
  printf("%s, %d CUs, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  run<0>("v_and_b32", out, p.multiProcessorCount);
  run<1>("v_bcnt_u32_b32", out, p.multiProcessorCount);
  run<2>("v_bitop3_b32", out, p.multiProcessorCount);
  run<3>("v_add_u32", out, p.multiProcessorCount);
  run<4>("v_lshrrev_b32", out, p.multiProcessorCount);
  run<5>("v_mov_b32_dpp", out, p.multiProcessorCount);
  run<6>("v_fma_f32", out, p.multiProcessorCount);
  run<7>("v_cndmask_b32 vcc", out, p.multiProcessorCount);
  run<8>("v_cndmask_b32 sgpr", out, p.multiProcessorCount);
  run<9>("v_alignbit_b32", out, p.multiProcessorCount);
  run<10>("v_lshrrev_b64", out, p.multiProcessorCount, 2.0);
  run<11>("v_add_u32_dpp", out, p.multiProcessorCount);
  return 0;
}
