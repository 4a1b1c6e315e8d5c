// Which XCD does workgroup b of a 1-D grid run on?  Reads HW_REG_XCC_ID (gfx940+) per workgroup and prints the
// distribution of (blockIdx.x % 8) per XCC id.  Build: hipcc -O2 --offload-arch=gfx950 scripts/xcc_probe.hip -o scripts/xcc_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned *out) {
  // s_getreg_b32: id 20 = HW_REG_XCC_ID, offset 0, width 4 -> simm16 = 20 | (0 << 6) | (3 << 11)
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
  if (threadIdx.x == 0) out[blockIdx.x] = xcc;
}
int main() {
  for (int grid : {2048, 1280, 256}) {
    unsigned *d; hipMalloc(&d, grid * 4);
    probe<<<grid, 256>>>(d);
    std::vector<unsigned> h(grid);
    hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost);
    int match = 0; int hist[16] = {0};
    for (int b = 0; b < grid; b++) { match += (h[b] == (unsigned)(b % 8)); hist[h[b] & 15]++; }
    printf("grid %d: xcc == blockIdx %% 8 for %d of %d workgroups; per-XCC counts:", grid, match, grid);
    for (int i = 0; i < 8; i++) printf(" %d", hist[i]);
    printf("\n");
    hipFree(d);
  }
  return 0;
}
