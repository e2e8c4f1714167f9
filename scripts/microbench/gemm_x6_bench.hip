// Microbenchmark of k_bt_gemm_x6 on one 512x512 layer at 65 536 rows (the shape that dominates the C3 training step),
// three epilogue modes.  Results and the phase-removal experiments made with it: profiles/r02_gemm_x6_microbench.md.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I generativerl_preview_amd/csrc -o gemm_x6_bench scripts/microbench/gemm_x6_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <string>
#include <cstdlib>
#include <cmath>
#include "plan.hpp"
namespace grlhip { inline int device_cus() { return 256; } }
#include "pack_f32.hip.inc"
#include "engine_batch.hip.inc"
namespace grlhip { namespace bt {
#include "gemm_x6_two_barrier.inc"
} }
using namespace grlhip;
int main(int argc, char** argv) {
  const int M = 512, K = 512; const int64_t ld = argc > 1 ? atoll(argv[1]) : 65536;
  float *A, *B, *C, *pre, *add, *bias;
  hipMalloc(&A, (size_t)K * M * 4); hipMalloc(&B, (size_t)K * ld * 4); hipMalloc(&C, (size_t)M * ld * 4);
  hipMalloc(&pre, (size_t)M * ld * 4); hipMalloc(&add, (size_t)M * ld * 4); hipMalloc(&bias, M * 4);
  hipMemset(A, 0, (size_t)K * M * 4); hipMemset(B, 0, (size_t)K * ld * 4); hipMemset(add, 0, (size_t)M * ld * 4); hipMemset(bias, 0, M * 4);
  std::vector<float> h((size_t)K * M); for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-4f;
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> hb((size_t)K * 4096); for (auto& v : hb) v = (rand() % 2001 - 1000) * 1e-3f;
  for (int64_t o = 0; o < K * ld; o += hb.size()) hipMemcpy(B + o, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  // the weights pre-split into their three bf16 pieces (what grlhip_pack_weights puts behind a GRLHIP_F32 image)
  __bf16* A3; hipMalloc(&A3, (size_t)(K / 16) * 3 * M * 16 * 2);
  {
    SplitBatch sb{};
    sb.a[0] = SplitArgs{A, A3, M, K, M, K / 16};
    sb.first_block[0] = 0; sb.first_block[1] = 64; sb.n = 1;
    hipLaunchKernelGGL(k_split_many, dim3(64), dim3(256), 0, 0, sb);
  }
  const int only_ns = argc > 2 ? atoi(argv[2]) : 0;   // 0: both
  hipFuncSetAttribute((const void*)bt::k_bt_gemm_x6<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bt::gemm_lds_bytes<2>());
  for (int ns = 2; ns <= 4; ++ns)   // 4: the pipelined kernel (16-k tiles, double-buffered LDS images)
  for (int mode = 0; mode < 3; ++mode) {
    if (only_ns && ns != only_ns) continue;
    bt::GemmArgs g{};
    g.A = A; g.bias = bias; g.B = B; g.C = C; g.M = M; g.K = K; g.lda = M; g.ld = ld; g.A3 = A3; g.np3 = M;
    g.pre = mode >= 1 ? pre : nullptr; g.add = mode >= 2 ? add : nullptr; g.flags = mode >= 1 ? (F_SILU_OUT | F_SAVE_PRE) : 0;
    const dim3 grid((unsigned)((ld / bt::BN) * ((M + bt::BM - 1) / bt::BM)));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
      if (ns == 2) hipLaunchKernelGGL((bt::k_bt_gemm_x6_two_barrier<false, 2, 2>), grid, dim3(256), 0, 0, g);
      else if (ns == 3) hipLaunchKernelGGL((bt::k_bt_gemm_x6_two_barrier<false, 2, 3>), grid, dim3(256), 0, 0, g);
      else hipLaunchKernelGGL((bt::k_bt_gemm_x6<false, 2>), grid, dim3(256), bt::gemm_lds_bytes<2>(), 0, g);
    };
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 100.0;
    printf("stages %d ld %lld mode %d (%s): %.1f us  %.1f TFLOP/s fp32-equivalent  %.2f TB/s min traffic\n", ns, (long long)ld, mode,
           mode == 0 ? "plain" : mode == 1 ? "silu+pre" : "silu+pre+add", us, 2.0 * M * K * ld / us * 1e-6,
           ((double)(K + M * (1 + (mode >= 1) + (mode >= 2))) * ld * 4) / us * 1e-6);
#ifdef GEMM_STAMPS
    if (ns == 4) {   // -DGEMM_STAMPS: where wave 0 of a workgroup spends its time (s_memtime ticks, mean over the workgroups)
      unsigned long long z[8] = {0};
      hipMemcpyToSymbol(HIP_SYMBOL(bt::gemm_dbg), z, sizeof z);
      launch();
      hipDeviceSynchronize();
      hipMemcpyFromSymbol(z, HIP_SYMBOL(bt::gemm_dbg), sizeof z);
      const double n = (double)z[6];
      printf("  per workgroup: prologue %.0f, k-loop %.0f (barrier wait %.0f, load wait %.0f), epilogue %.0f, whole %.0f ticks; MFMA time of the k-loop at 32 cycles each: %d\n",
             z[0] / n, z[1] / n, z[2] / n, z[3] / n, z[4] / n, z[5] / n, (K / 16) * 24 * 32);
    }
#endif
  }
  {  // the pipelined kernel accumulates the same products in the same order: bit-identical outputs expected
    bt::GemmArgs g{};
    g.A = A; g.bias = bias; g.B = B; g.C = C; g.M = M; g.K = K; g.lda = M; g.ld = ld; g.A3 = A3; g.np3 = M;
    g.pre = pre; g.add = add; g.flags = F_SILU_OUT | F_SAVE_PRE;
    const dim3 grid((unsigned)((ld / bt::BN) * ((M + bt::BM - 1) / bt::BM)));
    std::vector<float> c0((size_t)M * 4096), c1((size_t)M * 4096);
    hipLaunchKernelGGL((bt::k_bt_gemm_x6_two_barrier<false, 2, 2>), grid, dim3(256), 0, 0, g);
    hipMemcpy(c0.data(), C, c0.size() * 4, hipMemcpyDeviceToHost);
    hipMemset(C, 0, (size_t)M * ld * 4);
    hipLaunchKernelGGL((bt::k_bt_gemm_x6<false, 2>), grid, dim3(256), bt::gemm_lds_bytes<2>(), 0, g);
    hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0; double mx = 0;
    for (size_t i = 0; i < c0.size(); ++i) { if (c0[i] != c1[i]) ++bad; mx = std::max(mx, (double)fabsf(c0[i])); }
    printf("pipelined vs two-barrier kernel: %zu of %zu outputs differ (max |C| %.3f)\n", bad, c0.size(), mx);
  }
  return 0;
}
