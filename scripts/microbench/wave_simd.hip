// Which SIMD does wave w of a 512-thread (8-wave) workgroup run on?  HW_REG_HW_ID bits 5:4 = SIMD_ID, 11:8 = CU_ID.
// One workgroup per CU resident (120 KB of LDS requested), 1024 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 1) void k(unsigned* out) {
  extern __shared__ unsigned char lds[];
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
  if (threadIdx.x == 1000) lds[0] = 1;
}
int main() {
  unsigned* d; const int G = 1024;
  hipMalloc(&d, G * 8 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  k<<<G, 512, 120 * 1024>>>(d);
  static unsigned h[G * 8];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int pat[256] = {0};
  for (int b = 0; b < G; ++b) {
    int key = 0;
    for (int w = 0; w < 8; ++w) key |= ((h[b * 8 + w] >> 4) & 3) << (2 * w);
    // histogram of the SIMD sequence of waves 0..7, printed for the first few distinct ones
    static int seen[65536]; 
    if (!seen[key]++) { printf("wg %4d: simd of waves 0..7 =", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3); printf("\n"); }
  }
  int rr = 0, pairs = 0, other = 0;
  for (int b = 0; b < G; ++b) {
    int s[8]; for (int w = 0; w < 8; ++w) s[w] = (h[b * 8 + w] >> 4) & 3;
    bool isrr = true, ispairs = true;
    for (int w = 0; w < 8; ++w) { if (s[w] != s[w & 3]) isrr = false; }
    int cnt[4] = {0, 0, 0, 0}; for (int w = 0; w < 4; ++w) cnt[s[w]]++;
    for (int q = 0; q < 4; ++q) if (cnt[q] != 1) isrr = false;
    for (int w = 0; w < 8; w += 2) if (s[w] != s[w + 1]) ispairs = false;
    rr += isrr; pairs += ispairs && !isrr; other += !isrr && !ispairs;
  }
  printf("waves w and w+4 share a SIMD, waves 0-3 on four SIMDs: %d; waves 2i, 2i+1 share: %d; other: %d (of %d)\n", rr, pairs, other, G);
  return 0;
}
