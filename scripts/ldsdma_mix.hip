// Microbenchmark (r02): does an HBM-missing LDS-DMA stream slow down L2-hitting LDS-DMA streams of the SAME CU?
// The 16-bit GEMM fills its LDS at ~27 GB/s per CU whatever the loop structure (r02 calls 2-5) although 75-90 % of its requests hit
// the L2 and the fabric carries < 3 TB/s.  Hypothesis: completion is in order per wave (vmcnt) and the CU has a bounded number of
// outstanding requests, so the first-touch misses of the A operand (HBM latency) set the pace of every piece queued behind them.
//
// extern "C" int ldsdma_mix(int mode, int miss_every, int iters, double* gbps_hit, double* gbps_miss, double* ms)
//   8 waves per block, one block per CU, every wave keeps 8 pieces of 1 KiB in flight (wait for the previous batch only).
//   mode 0: every wave streams the shared L2-resident window (1 MB per XCD)                        -> all hits
//   mode 1: every wave: piece j of a batch comes from its PRIVATE HBM window if j % miss_every == 0  -> misses mixed into every queue
//   mode 2: waves 0..1 stream only their private HBM windows, waves 2..7 only the shared window        -> misses in dedicated waves
//   returns per-class GB/s per CU (bytes of that class / the slowest wave's time of that class)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#define DMA_PIECE(GSRC, LDSB)                                                                                 \
  {                                                                                                           \
    unsigned keep_;                                                                                           \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(GSRC), "s"(LDSB) : "memory");                                           \
  }

__global__ __launch_bounds__(512) void mix_kernel(const char* __restrict__ shared_win, const char* __restrict__ priv, size_t priv_per_wave,
                                                  int mode, int miss_every, int iters, unsigned long long* __restrict__ cycles) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * (16u * 1024u));
  const char* sh = shared_win + (size_t)(blockIdx.x & 7) * (1u << 20) + (size_t)lane * 16;          // 1 MB window per XCD
  const char* pv = priv + ((size_t)blockIdx.x * 8 + wave) * priv_per_wave + (size_t)lane * 16;       // private stream of this wave
  size_t soff = (size_t)wave * 131072, poff = 0;        // every wave walks its own 128 KB slice of the XCD's shared window
  const bool miss_wave = mode == 2 && wave < 2;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const unsigned half = (unsigned)((it & 1) * 8192);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool miss = mode == 1 ? (j % miss_every == 0) : miss_wave;
      const char* g;
      if (miss) { g = pv + poff; poff += 1024; if (poff + 1024 > priv_per_wave) poff = 0; }
      else { g = sh + soff; soff += 1024; if (soff >= (size_t)(wave + 1) * 131072) soff = (size_t)wave * 131072; }
      DMA_PIECE(g, lds_wave + half + (unsigned)(j * 1024))
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
  __syncthreads();
  if (threadIdx.x == 0 && smem[17] == 255 && iters < 0) cycles[0] = 0;
}

extern "C" int ldsdma_mix(int mode, int miss_every, int iters, double* gbps_hit, double* gbps_miss, double* ms_out) {
  const int blocks = 256;
  const size_t priv_per_wave = (size_t)8 << 20;                  // 8 MB per wave -> 16 GB of private streams: HBM, never re-read
  char *sh, *pv; unsigned long long* cyc;
  if (hipMalloc((void**)&sh, 8u << 20) != hipSuccess) return -1;
  if (hipMalloc((void**)&pv, priv_per_wave * 8 * blocks) != hipSuccess) return -2;
  (void)hipMalloc((void**)&cyc, blocks * 8 * sizeof(unsigned long long));
  (void)hipMemset(sh, 1, 8u << 20); (void)hipMemset(pv, 2, priv_per_wave * 8 * blocks);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mix_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(mix_kernel, dim3(blocks), dim3(512), 128 * 1024, 0, sh, pv, priv_per_wave, mode, miss_every, iters / 4 + 1, cyc);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(mix_kernel, dim3(blocks), dim3(512), 128 * 1024, 0, sh, pv, priv_per_wave, mode, miss_every, iters, cyc);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
  *ms_out = ms;
  // per-class bytes per CU over the kernel time (classes finish at different times in mode 2: use each class's own slowest wave,
  // scaled from cycles to time through the kernel's duration / the slowest wave overall)
  std::vector<unsigned long long> h(blocks * 8);
  (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  unsigned long long cmax = 1, cmax_hit = 1, cmax_miss = 1;
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < 8; ++w) {
      const unsigned long long c = h[b * 8 + w];
      cmax = c > cmax ? c : cmax;
      const bool mw = mode == 2 && w < 2;
      if (mw) cmax_miss = c > cmax_miss ? c : cmax_miss; else cmax_hit = c > cmax_hit ? c : cmax_hit;
    }
  const double sec_per_cycle = (ms * 1e-3) / (double)cmax;
  double hit_bytes, miss_bytes, t_hit, t_miss;
  if (mode == 2) {
    hit_bytes = 6.0 * 8 * 1024.0 * iters; miss_bytes = 2.0 * 8 * 1024.0 * iters;
    t_hit = cmax_hit * sec_per_cycle; t_miss = cmax_miss * sec_per_cycle;
  } else {
    const int nmiss = mode == 1 ? (8 + miss_every - 1) / miss_every : 0;
    hit_bytes = 8.0 * (8 - nmiss) * 1024.0 * iters; miss_bytes = 8.0 * nmiss * 1024.0 * iters;
    t_hit = t_miss = ms * 1e-3;
  }
  *gbps_hit = hit_bytes / t_hit / 1e9; *gbps_miss = miss_bytes / t_miss / 1e9;
  (void)hipFree(sh); (void)hipFree(pv); (void)hipFree(cyc);
  return 0;
}
