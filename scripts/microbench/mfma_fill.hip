// Microbenchmark: which filler instructions hide in the shadow of a dependent v_mfma_f32_32x32x16_bf16 chain
// (one wave per SIMD).  Each gap = 1 MFMA + a filler group written in inline asm so the order is exact.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(int iters, float* out, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63;
  f32x16 acc = {0};
  f32x16 other = {0};   // unrelated accumulator (AGPR): source of v_accvgpr_read fillers
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (lane + e)); b[e] = (__bf16)(0.02f * (lane - e)); }
  float x0 = lane * 0.1f, x1 = 1.0f, x2 = 2.0f, x3 = 3.0f, y0 = 0.f, y1 = 0.f;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 z0 = {0.f, 0.f}, z1 = {1.f, 2.f};
  asm volatile("" : "+a"(other));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma clang loop unroll(disable)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (MODE == 5) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
      if (MODE == 1)  // 4 plain VALU (16 issue cycles)
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
      if (MODE == 2)  // mul + exp + add + rcp (24 issue cycles)
        asm volatile("v_mul_f32 %0, %0, %1\n v_exp_f32 %1, %0\n v_add_f32 %2, 1.0, %1\n v_rcp_f32 %3, %2" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
      if (MODE == 3)  // 2 accvgpr reads of an unrelated accumulator + 2 VALU
        asm volatile("v_accvgpr_read_b32 %0, %4\n v_accvgpr_read_b32 %1, %4\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "a"(other[m & 15]));
      if (MODE == 4)  // 2 accvgpr writes + 2 VALU
        asm volatile("v_accvgpr_write_b32 %4, %0\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1\n v_fma_f32 %1, %1, %0, %2" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+a"(other[m & 15]));
      if (MODE >= 10) {  // a full gap: 4 fma + two more instructions of the kind under test (independent)
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        if (MODE == 10) asm volatile("v_fma_f32 %0, %2, %2, %2\n v_fma_f32 %1, %3, %3, %3" : "=v"(y0), "=v"(y1) : "v"(x2), "v"(x3));
        if (MODE == 11) asm volatile("v_exp_f32 %0, %2\n v_exp_f32 %1, %3" : "=v"(y0), "=v"(y1) : "v"(x2), "v"(x3));
        if (MODE == 12) asm volatile("v_rcp_f32 %0, %2\n v_rcp_f32 %1, %3" : "=v"(y0), "=v"(y1) : "v"(x2), "v"(x3));
        if (MODE == 13) asm volatile("v_exp_f16 %0, %2\n v_exp_f16 %1, %3" : "=v"(y0), "=v"(y1) : "v"(x2), "v"(x3));
        if (MODE == 14) asm volatile("v_rcp_f16 %0, %2\n v_rcp_f16 %1, %3" : "=v"(y0), "=v"(y1) : "v"(x2), "v"(x3));
        if (MODE == 15) asm volatile("v_exp_f32 %0, %2\n v_rcp_f32 %1, %3" : "=v"(y0), "=v"(y1) : "v"(x2), "v"(x3));
        if (MODE == 16) asm volatile("v_exp_f32 %0, %1" : "=v"(y0) : "v"(x2));
        if (MODE == 17) asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(z0) : "v"(z1));
        if (MODE == 18) asm volatile("v_rcp_iflag_f32 %0, %2\n v_rsq_f32 %1, %3" : "=v"(y0), "=v"(y1) : "v"(x2), "v"(x3));
      }
      if (MODE == 6)  // cvt_pk + 3 VALU
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %1\n v_fma_f32 %3, %3, %1, %2" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = x0 + x1 + x2 + x3 + y0 + y1 + z0[0] + z0[1];
  for (int r = 0; r < 16; ++r) s += acc[r] + other[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(float* out, unsigned long long* dc, const char* what) {
  const int iters = 20000;
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, 200, out, dc);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, iters, out, dc);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), dc, 256 * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += (double)v; avg /= 256;
  printf("mode %d (%s): %.1f cycles per MFMA gap\n", MODE, what, avg / iters / 16);
}

int main() {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  unsigned long long* dc; hipMalloc(&dc, 256 * 8);
  run<0>(out, dc, "MFMA only, B in VGPR");
  run<5>(out, dc, "MFMA only, B in AGPR");
  run<1>(out, dc, "4 x v_fma");
  run<2>(out, dc, "mul exp add rcp");
  run<3>(out, dc, "2 accvgpr_read (other acc) + 2 fma");
  run<4>(out, dc, "1 accvgpr_write (other acc) + 3 fma");
  run<6>(out, dc, "cvt_pk + 3 fma");
  run<10>(out, dc, "4 fma + 2 fma");
  run<11>(out, dc, "4 fma + 2 v_exp_f32");
  run<12>(out, dc, "4 fma + 2 v_rcp_f32");
  run<13>(out, dc, "4 fma + 2 v_exp_f16");
  run<14>(out, dc, "4 fma + 2 v_rcp_f16");
  run<15>(out, dc, "4 fma + v_exp_f32 + v_rcp_f32");
  run<16>(out, dc, "4 fma + 1 v_exp_f32");
  run<17>(out, dc, "4 fma + 1 v_pk_mul_f32");
  run<18>(out, dc, "4 fma + v_rcp_iflag_f32 + v_rsq_f32");
  return 0;
}
