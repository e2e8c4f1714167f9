// dW (weight gradient) kernels of the fp32 batch engine on one N x K layer over `ld` rows: k_bt_dw_x6 (the library's: both
// kinds of work in every wave, two workgroups per CU) against k_bt_dw_ps (dw_ps_kernel.inc, measured and not kept: staging
// waves + matrix waves, one workgroup per CU), time and bit-identity of the partials; with -DDW_STAMPS the phase cycles
// of k_bt_dw_ps.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=1048576 -I generativerl_preview_amd/csrc -I include -I scripts/microbench
//         [-DDW_STAMPS] scripts/microbench/dw_bench.hip generativerl_preview_amd/csrc/plan.cpp -o scripts/microbench/dw_bench
//   dw_bench [N] [K] [ld]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <string>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <type_traits>
#include "plan.hpp"
namespace grlhip { inline int device_cus() { return 256; } }
#include "pack_f32.hip.inc"
#include "engine_batch.hip.inc"
namespace grlhip { namespace bt {
#include "dw_ps_kernel.inc"
} }
using namespace grlhip;
#ifndef NSW
#define NSW bt::DW_NSW
#endif

template <typename F>
static float time_us(F f, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 512, K = argc > 2 ? atoi(argv[2]) : 512;
  const int64_t ld = argc > 3 ? atoll(argv[3]) : 65536;
  std::vector<float> h((size_t)std::max(N, K) * ld);
  uint32_t x = 12345u;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
  float *dZ, *X, *part[2];
  hipMalloc(&dZ, (size_t)N * ld * 4); hipMalloc(&X, (size_t)K * ld * 4);
  hipMemcpy(dZ, h.data(), (size_t)N * ld * 4, hipMemcpyHostToDevice);
  std::reverse(h.begin(), h.end());
  hipMemcpy(X, h.data(), (size_t)K * ld * 4, hipMemcpyHostToDevice);
  bt::DwArgs d{};
  d.dZ = dZ; d.X = X; d.N = N; d.K = K; d.ld = ld; d.gw = nullptr;
  const int tm = (N + bt::BM - 1) / bt::BM, tn = (K + bt::BN - 1) / bt::BN;
  const int64_t splits = dw_splits(N, K, ld, &d.tiles_per_split);
  const size_t pf = (size_t)splits * ((size_t)N * K + N);
  for (int i = 0; i < 2; ++i) { hipMalloc(&part[i], pf * 4 + 4096); hipMemset(part[i], 0, pf * 4); }
  d.gb = part[0];   // (only "is there a bias": non-null)
  gemm_lds_opt_in();
  hipFuncSetAttribute((const void*)bt::k_bt_dw_ps<3, NSW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bt::dw_ps_lds_bytes<3>());
  hipFuncSetAttribute((const void*)bt::k_bt_dw_ps<2, NSW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bt::dw_ps_lds_bytes<2>());
  const dim3 grid(tn, tm, (unsigned)splits);
  printf("dW %d x %d over %lld rows: grid %d x %d x %lld, %d row tiles per slice\n", N, K, (long long)ld, tn, tm, (long long)splits, d.tiles_per_split);
  const double gflop = 2.0 * N * K * (double)ld * 1e-9;
  for (int np = 3; np >= 2; --np) {
    bt::DwArgs a = d, b = d; a.part = part[0]; b.part = part[1];
    const float t_old = time_us([&] { if (np == 3) hipLaunchKernelGGL(bt::k_bt_dw_x6<3>, grid, dim3(256), 0, 0, a); else hipLaunchKernelGGL(bt::k_bt_dw_x6<2>, grid, dim3(256), 0, 0, a); }, 20);
    const float t_new = time_us([&] { if (np == 3) hipLaunchKernelGGL((bt::k_bt_dw_ps<3, NSW>), grid, dim3(256 + 64 * NSW), bt::dw_ps_lds_bytes<3>(), 0, b); else hipLaunchKernelGGL((bt::k_bt_dw_ps<2, NSW>), grid, dim3(256 + 64 * NSW), bt::dw_ps_lds_bytes<2>(), 0, b); }, 20);
    std::vector<float> pa(pf), pb(pf);
    hipMemcpy(pa.data(), part[0], pf * 4, hipMemcpyDeviceToHost); hipMemcpy(pb.data(), part[1], pf * 4, hipMemcpyDeviceToHost);
    const bool same = memcmp(pa.data(), pb.data(), pf * 4) == 0;
    const int terms = np == 3 ? 6 : 3;
    printf("pieces %d: k_bt_dw_x6 %7.1f us (%5.0f TFLOP/s bf16)   k_bt_dw_ps %7.1f us (%5.0f TFLOP/s bf16)   partials %s\n", np, t_old,
           gflop * terms / t_old * 1e-3, t_new, gflop * terms / t_new * 1e-3, same ? "bit-identical" : "DIFFER");
#ifdef DW_STAMPS
    unsigned long long z[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(bt::dw_dbg), z, sizeof z);
    if (np == 3) hipLaunchKernelGGL((bt::k_bt_dw_ps<3, NSW>), grid, dim3(256 + 64 * NSW), bt::dw_ps_lds_bytes<3>(), 0, b); else hipLaunchKernelGGL((bt::k_bt_dw_ps<2, NSW>), grid, dim3(256 + 64 * NSW), bt::dw_ps_lds_bytes<2>(), 0, b);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(z, HIP_SYMBOL(bt::dw_dbg), sizeof z);
    const double wg = (double)tn * tm * splits, tiles = wg * d.tiles_per_split;
    printf("  s_memtime ticks per tile: staging load wait %.0f, split+stores+requests %.0f, staging barrier %.0f | matrix tile %.0f, matrix barrier %.0f | per workgroup: whole %.0f, epilogue %.0f\n",
           z[0] / tiles, z[1] / tiles, z[2] / tiles, z[3] / tiles, z[4] / tiles, z[5] / wg, z[6] / wg);
#endif
  }
  return 0;
}
