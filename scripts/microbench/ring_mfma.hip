// Microbenchmark: the bf16 sampler's weight ring in isolation -- 4 waves, LDS-DMA refill, one s_barrier per
// chunk, P-deep register prefetch queue, one dependent MFMA chain per wave and nothing else.
// Build variants with -DGRLHIP_X_NOLOAD / -DGRLHIP_X_NORING / -DGRLHIP_X_NOWAIT (see Ring::advance).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../generativerl_preview_amd/csrc/plan.hpp"
#include "../../generativerl_preview_amd/csrc/engine_bf16.hip.inc"
using namespace grlhip::b16;

template <int NSLOT, int CF, int P, int CHUNKS, int ACCS>
__global__ __launch_bounds__(256, 1) void k(const char* stream, int iters, float* out, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(1024))) char lds[NSLOT * CF * FRAG_BYTES];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  using RingT = Ring<NSLOT, CF, P, CHUNKS, CHUNKS>;
  RingT ring;
  ring.init(stream, lds, wave, lane);
  ring.start();
  f32x16 acc[ACCS];
  for (int i = 0; i < ACCS; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 b;
  for (int e = 0; e < 8; ++e) b[e] = (__bf16)(0.01f * (lane + e));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma clang loop unroll(disable)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < RingT::STEP; ++i) {
      acc[i % ACCS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring.get(i), b, acc[i % ACCS], 0, 0, 0);
#if defined(SGB)
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one DS read (the refill issued by get(i)) ...
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // ... then one MFMA
#endif
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  ring.drain();
  float s = 0;
  for (int i = 0; i < ACCS; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NSLOT, int CF, int P, int CHUNKS, int ACCS>
void run(const char* d, float* out, unsigned long long* dc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<NSLOT, CF, P, CHUNKS, ACCS>), dim3(256), dim3(256), 0, 0, d, 50, out, dc);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<NSLOT, CF, P, CHUNKS, ACCS>), dim3(256), dim3(256), 0, 0, d, iters, out, dc);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), dc, 256 * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += (double)v; avg /= 256;
  printf("NSLOT %d CF %2d P %2d chunks %3d accs %d: %.1f cycles per MFMA\n", NSLOT, CF, P, CHUNKS, ACCS, avg / iters / (CHUNKS * CF));
}

int main() {
  char* d; hipMalloc(&d, 4 << 20); hipMemset(d, 0, 4 << 20);
  float* out; hipMalloc(&out, 256 * 256 * 4);
  unsigned long long* dc; hipMalloc(&dc, 256 * 8);
  run<4, 8, 8, 16, 1>(d, out, dc);
  run<4, 16, 8, 8, 1>(d, out, dc);
  run<4, 32, 8, 4, 1>(d, out, dc);
  run<4, 16, 8, 200, 1>(d, out, dc);   // 3.1 MiB stream like the sampler
  run<4, 32, 8, 100, 1>(d, out, dc);
  run<4, 32, 16, 100, 1>(d, out, dc);
  return 0;
}
