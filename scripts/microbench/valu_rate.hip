// Issue rate of the VALU instructions of the fp32 -> bf16 split (gfx950), alone and beside an MFMA wave on the same SIMD.
// One workgroup per CU; every wave runs LOOPS x 32 independent instructions of one kind; cycles from s_memtime of wave 0.
//   mode A: NW waves per SIMD all run the VALU loop (NW = 1, 2)
//   mode B: one wave per SIMD runs the VALU loop, a second wave on the same SIMD issues independent MFMAs back to back
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define REP32(X) X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X
template <int OP>
__device__ __forceinline__ void valu32(float (&r)[8], unsigned (&u)[8]) {
  // 32 instructions, 8 independent chains of 4 (every instruction depends on the one 8 before it)
#define I(k) \
  if (OP == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[k]) : "v"(r[k]), "v"(r[(k + 1) & 7])); \
  else if (OP == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*(double*)&r[(k & 3) * 2]) : "v"(*(double*)&r[(k & 3) * 2]), "v"(*(double*)&r[((k + 1) & 3) * 2])); \
  else if (OP == 2) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[k]) : "v"(u[k])); \
  else if (OP == 3) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[k]) : "v"(u[k])); \
  else if (OP == 4) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r[k]) : "v"(r[k]), "v"(r[(k + 1) & 7])); \
  else if (OP == 5) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(r[k]) : "v"(u[k]), "v"(u[(k + 1) & 7])); \
  else if (OP == 6) asm volatile("v_cndmask_b32 %0, 0, %1, vcc" : "=v"(r[k]) : "v"(r[k]) : );
  I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
  I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#undef I
}
template <int OP, int MODE>
__global__ __launch_bounds__(512, 1) void k(unsigned long long* out, float* sink, int loops) {
  extern __shared__ unsigned char lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float r[8]; unsigned u[8];
  for (int i = 0; i < 8; ++i) { r[i] = threadIdx.x * 0.001f + i; u[i] = threadIdx.x * 77u + i; }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 1 && wave >= 4) {
    f32x16 acc[4]; bf16x8 a, b;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)1.0f; b[e] = (__bf16)0.5f; }
    for (int l = 0; l < loops; ++l)   // 8 independent-enough MFMAs per iteration (4 accumulators)
      for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0];
    if (s == 123.f) sink[0] = s;
  } else if (wave < 4 || MODE == 0) {
    for (int l = 0; l < loops; ++l) valu32<OP>(r, u);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 8; ++i) s += r[i] + (float)u[i];
  if (s == 123.456f) sink[1] = s;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[wave] = t1 - t0;
  if (threadIdx.x == 1000) lds[0] = 1;
}
template <int OP>
void run(const char* name, unsigned long long* d, float* sink) {
  const int loops = 2000;
  unsigned long long h[8];
  double res[3];
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int threads = cfg == 0 ? 256 : 512;
    if (cfg < 2) { hipFuncSetAttribute((const void*)k<OP, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024); k<OP, 0><<<256, threads, 120 * 1024>>>(d, sink, loops); }
    else { hipFuncSetAttribute((const void*)k<OP, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024); k<OP, 1><<<256, threads, 120 * 1024>>>(d, sink, loops); }
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    res[cfg] = (double)h[0] / (loops * 32.0);
    if (cfg == 2) printf("%-22s cycles per instruction: alone %.2f | two VALU waves per SIMD %.2f (per wave) | beside an MFMA wave %.2f (the MFMA wave: %.1f cycles per MFMA)\n",
                         name, res[0], res[1], res[2], (double)h[4] / (loops * 8.0));
  }
}
int main() {
  unsigned long long* d; float* sink;
  hipMalloc(&d, 64); hipMalloc(&sink, 64);
  run<0>("v_cvt_pk_bf16_f32", d, sink);
  run<1>("v_pk_add_f32", d, sink);
  run<2>("v_lshlrev_b32", d, sink);
  run<3>("v_and_b32", d, sink);
  run<4>("v_sub_f32", d, sink);
  run<5>("v_dot2c_f32_bf16", d, sink);
  run<6>("v_cndmask_b32", d, sink);
  return 0;
}
