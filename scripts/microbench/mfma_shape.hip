// Microbenchmark: the two bf16 MFMA shapes in an LDS-fed loop shaped like k_sample_bf16's (one wave per SIMD, every A
// operand a ds_read_b128 of a 1 KiB weight fragment, B operands in registers, random data).  Per iteration and wave:
//   SHAPE 0: 16 x v_mfma_f32_32x32x16_bf16, 16 fragment reads              (524 288 FLOP)
//   SHAPE 1: 32 x v_mfma_f32_16x16x32_bf16, 16 fragment reads, each fragment feeding two MFMAs (two 16-row B tiles)
// Reports shader cycles (s_memtime) AND wall time per iteration: the clock the chip holds differs between the shapes
// (MI355X_MICROARCH.md, DVFS give-back item 7).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void k(const unsigned* seed, int iters, float* out, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(1024))) char lds[64 * 1024];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += 256) ((unsigned*)lds)[i] = seed[i];   // random bf16 pairs
  __syncthreads();
  bf16x8 b0, b1;
  for (int e = 0; e < 8; ++e) { b0[e] = (__bf16)(0.01f * ((lane * 7 + e * 3) % 61) - 0.3f); b1[e] = (__bf16)(0.02f * ((lane * 5 + e) % 53) - 0.5f); }
  const char* rd = lds + lane * 16;
  f32x16 acc0 = {0}, acc1 = {0};
  f32x4 c[8] = {};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  // fragments are read one iteration ahead (register double buffer), so that the loop is bound by the MFMAs
  bf16x8 fa[16], fb[16];
  auto fetch = [&](bf16x8 (&f)[16], int it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) f[m] = *(const bf16x8*)(rd + ((it & 3) * 16 + m) * 1024);
  };
  auto mul = [&](const bf16x8 (&f)[16]) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (SHAPE == 0) {
        if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[m], b1, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[m], b0, acc0, 0, 0, 0);
      } else {
        c[(m & 3) * 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[m], b0, c[(m & 3) * 2], 0, 0, 0);
        c[(m & 3) * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[m], b1, c[(m & 3) * 2 + 1], 0, 0, 0);
      }
    }
  };
  fetch(fa, 0);
#pragma clang loop unroll(disable)
  for (int it = 0; it < iters; it += 2) {
    fetch(fb, it + 1);
    mul(fa);
    fetch(fa, it + 2);
    mul(fb);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  for (int q = 0; q < 8; ++q) s += c[q][0] + c[q][1] + c[q][2] + c[q][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE>
void run(const unsigned* seed, float* out, unsigned long long* dc) {
  const int iters = 200000;
  hipLaunchKernelGGL((k<SHAPE>), dim3(256), dim3(256), 0, 0, seed, 2000, out, dc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<SHAPE>), dim3(256), dim3(256), 0, 0, seed, iters, out, dc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), dc, 256 * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += (double)v; avg /= 256;
  const double flop = 256.0 * 4 * iters * 16 * 32768.0;
  printf("%s: %.1f cycles per iteration (512 ideal), %.3f us per iteration, clock %.2f GHz, %.0f TFLOP/s\n",
         SHAPE == 0 ? "32x32x16" : "16x16x32", avg / iters, ms * 1e3 / iters, avg / (ms * 1e6), flop / (ms * 1e-3) * 1e-12);
}

int main() {
  std::vector<unsigned> h(16384);
  unsigned x = 12345u;
  for (auto& v : h) {   // two random bf16 values in [-1, 1) per dword
    x = x * 1664525u + 1013904223u; const float a = (int)(x >> 8 & 0xffff) / 32768.0f - 1.0f;
    x = x * 1664525u + 1013904223u; const float b = (int)(x >> 8 & 0xffff) / 32768.0f - 1.0f;
    unsigned ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
    v = (ua >> 16) | (ub & 0xffff0000u);
  }
  unsigned* seed; hipMalloc(&seed, h.size() * 4); hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  float* out; hipMalloc(&out, 256 * 256 * 4);
  unsigned long long* dc; hipMalloc(&dc, 256 * 8);
  for (int rep = 0; rep < 2; ++rep) { run<0>(seed, out, dc); run<1>(seed, out, dc); }
  return 0;
}
