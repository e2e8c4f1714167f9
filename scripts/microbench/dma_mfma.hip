// Microbenchmark: cost of LDS-DMA pieces issued inside an MFMA stream (one wave per SIMD, 4 waves per CU).
// Per iteration: 16 x v_mfma_f32_32x32x16_bf16 (A from LDS via ds_read_b128, B in registers) and NP DMA pieces,
// either clustered at the start of the iteration (MODE 0) or one piece every 16/NP MFMAs (MODE 1);
// MODE 2: A operands stay in registers (no LDS reads).
// MODE 3: like MODE 1, but the refill goes through registers: buffer_load_dwordx4 -> VGPRs, and one iteration later
// ds_write_b128 -> LDS (4 staging quads per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PIECE(p, slotoff, choff) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + (slotoff)), 16, voff, (choff), (p) * 1024, 0)

template <int NP, int MODE>
__global__ __launch_bounds__(256, 1) void k(const char* stream, int iters, float* out, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(1024))) char lds[64 * 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)stream, 0, 0x7fffffff, 0x00020000);
  const int voff = lane * 16 + wave * 4096;
  for (int i = threadIdx.x; i < 16384; i += 256) ((float*)lds)[i] = 0.001f * (i & 63);
  __syncthreads();
  f32x16 acc0 = {0}, acc1 = {0};
  bf16x8 b;
  for (int e = 0; e < 8; ++e) b[e] = (__bf16)(0.01f * (lane + e));
  bf16x8 areg = b;
  const char* rd = lds + lane * 16;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 stg[4] = {};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma clang loop unroll(disable)
  for (int it = 0; it < iters; ++it) {
    const int slot = (it & 3) * 16384, ch = (it % 200) * 16384;
    if (NP > 0 && MODE != 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    if (MODE == 0) {
      if (NP >= 1) PIECE(0, slot + wave * 4096, ch);
      if (NP >= 2) PIECE(1, slot + wave * 4096, ch);
      if (NP >= 3) PIECE(2, slot + wave * 4096, ch);
      if (NP >= 4) PIECE(3, slot + wave * 4096, ch);
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      bf16x8 a = MODE == 2 ? areg : *(const bf16x8*)(rd + ((it + 1) & 3) * 16384 + m * 1024);
      if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      if (MODE == 3) {
        // write the piece loaded one iteration ago, then reload its staging quad
        char* wr = lds + slot + wave * 4096 + lane * 16;
#define RT(q, mm) if (m == (mm) && NP > (q)) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP - 1) : "memory"); \
          *(i32x4*)(wr + (q) * 1024) = stg[q]; \
          stg[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (q) * 1024, ch, 0); }
        RT(0, 1) RT(1, 5) RT(2, 9) RT(3, 13)
      } else if (MODE != 0 && NP > 0) {
        if (m == 1 && NP >= 1) PIECE(0, slot + wave * 4096, ch);
        if (m == 5 && NP >= 2) PIECE(1, slot + wave * 4096, ch);
        if (m == 9 && NP >= 3) PIECE(2, slot + wave * 4096, ch);
        if (m == 13 && NP >= 4) PIECE(3, slot + wave * 4096, ch);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NP, int MODE>
void run(const char* d, float* out, unsigned long long* dc) {
  const int iters = 20000;
  hipLaunchKernelGGL((k<NP, MODE>), dim3(256), dim3(256), 0, 0, d, 200, out, dc);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<NP, MODE>), dim3(256), dim3(256), 0, 0, d, iters, out, dc);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), dc, 256 * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += (double)v; avg /= 256;
  printf("pieces/16 MFMA %d mode %d: %.1f cycles per iteration (16 MFMA = 512 ideal), %.1f per MFMA\n", NP, MODE, avg / iters, avg / iters / 16);
}

int main() {
  char* d; hipMalloc(&d, (size_t)200 * 16384 + 65536); hipMemset(d, 0, (size_t)200 * 16384 + 65536);
  float* out; hipMalloc(&out, 256 * 256 * 4);
  unsigned long long* dc; hipMalloc(&dc, 256 * 8);
  run<0, 0>(d, out, dc); run<1, 0>(d, out, dc); run<2, 0>(d, out, dc); run<4, 0>(d, out, dc);
  run<1, 1>(d, out, dc); run<2, 1>(d, out, dc); run<4, 1>(d, out, dc);
  run<0, 2>(d, out, dc); run<2, 2>(d, out, dc); run<4, 2>(d, out, dc);
  run<2, 3>(d, out, dc); run<4, 3>(d, out, dc);
  return 0;
}
