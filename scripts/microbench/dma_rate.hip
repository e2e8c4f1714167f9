// Microbenchmark: sustained L2 -> LDS rate of LDS-DMA (buffer_load_dwordx4 ... lds) per CU, 4 waves per CU,
// streaming a 3.25 MiB L2-resident image repeatedly (the bf16 sampler's weight stream pattern).
// Build: hipcc -O3 --offload-arch=gfx950 dma_rate.hip -o dma_rate ; run: ./dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int WAVES_ISSUING>
__global__ __launch_bounds__(256, 1) void k_dma(const char* stream, int chunks, int reps, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(1024))) char lds[64 * 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)stream, 0, 0x7fffffff, 0x00020000);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < WAVES_ISSUING) {
    constexpr int PER = 4 / WAVES_ISSUING;  // pieces of a 16 KiB chunk quarter this wave issues (x4 pieces each)
    for (int r = 0; r < reps; ++r) {
      for (int c = 0; c < chunks; ++c) {
        char* d = lds + (c & 3) * 16384;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          const int quarter = wave * PER + q;
#define PIECE(p) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(d + quarter * 4096), 16, \
                                                     lane * 16 + quarter * 4096, c * 16384, p * 1024, 0)
          PIECE(0); PIECE(1); PIECE(2); PIECE(3);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PER) : "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int W>
void run(const char* d, int chunks, int reps, unsigned long long* dc) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_dma<W>, dim3(256), dim3(256), 0, 0, d, chunks, 2, dc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_dma<W>, dim3(256), dim3(256), 0, 0, d, chunks, reps, dc);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), dc, 256 * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += (double)v; avg /= 256;
  const double bytes = (double)chunks * 16384 * reps;
  printf("waves issuing %d: %.3f ms, %.0f cycles/CU, %.1f B/clk/CU, %.1f cycles per 1 KiB piece, %.2f TB/s aggregate\n", W, ms, avg,
         bytes / avg, avg / (bytes / 1024), bytes * 256 / (ms * 1e-3) / 1e12);
}

int main() {
  const int chunks = 208;  // 3.25 MiB
  char* d; hipMalloc(&d, (size_t)chunks * 16384); hipMemset(d, 1, (size_t)chunks * 16384);
  unsigned long long* dc; hipMalloc(&dc, 256 * 8);
  run<4>(d, chunks, 50, dc);
  run<2>(d, chunks, 50, dc);
  run<1>(d, chunks, 50, dc);
  return 0;
}
