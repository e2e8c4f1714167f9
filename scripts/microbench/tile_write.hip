// How fast can 256 KB tiles of a [1024 features][262144 rows] fp32 matrix be written, by tile shape?  (The epilogue of the
// 256 x 256-tile GEMM writes 256 row segments of 1 KB, 1 MB apart; torch's fill_ writes the same 1 GiB at 6.6 TB/s.)
//   hipcc -O3 --offload-arch=gfx950 -o tile_write tile_write.hip && ./tile_write
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int64_t M = 1024, LD = 262144;
template <int TR>   // rows per tile; features per tile = 65536 / TR
__global__ __launch_bounds__(512) void k_write(float* C, int order) {
  constexpr int TF = 65536 / TR;
  const int64_t ntc = LD / TR; const int mtc = M / TF;
  int64_t L = blockIdx.x, nt; int mt;
  if (order == 0) { const int64_t slot = L >> 3; const int xcd = L & 7; const int64_t g = slot / mtc; nt = g * 8 + xcd; mt = slot % mtc; }
  else { nt = L % ntc; mt = L / ntc; }
  if (nt >= ntc) return;
  const int tid = threadIdx.x;
  constexpr int QPR = TR / 4;                     // float4 per tile row
  for (int i = tid; i < TF * QPR; i += 512) {
    const int f = i / QPR, q = i % QPR;
    *(f32x4*)(C + (size_t)(mt * TF + f) * LD + nt * TR + q * 4) = f32x4{1.f, 2.f, 3.f, (float)i};
  }
}
template <int TR> void run(float* C, int order) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const unsigned grid = (unsigned)((M * LD) / 65536);
  hipLaunchKernelGGL(k_write<TR>, dim3(grid), dim3(512), 0, 0, C, order);
  hipEventRecord(a);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_write<TR>, dim3(grid), dim3(512), 0, 0, C, order);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("tile %5d features x %5d rows (segments of %6d B), order %d: %7.1f us  %.2f TB/s\n", 65536 / TR, TR, TR * 4, order, ms * 100, M * LD * 4 / (ms / 10 * 1e-3) / 1e12);
}
int main() {
  float* C; hipMalloc(&C, M * LD * 4);
  for (int order = 0; order < 2; ++order) { run<64>(C, order); run<256>(C, order); run<1024>(C, order); run<4096>(C, order); run<16384>(C, order); }
  return 0;
}
