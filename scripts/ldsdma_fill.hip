// Microbenchmark: how fast can one CU / the whole chip stream global memory into the LDS with global_load_lds_dwordx4 (the 16-bit
// GEMM's tile transport), as a function of (a) waves issuing per CU, (b) 1 KiB pieces in flight per wave, (c) whether a batch is
// drained before the next one is issued (what a 2-stage k-loop does: s_waitcnt vmcnt(0) + barrier per k-tile) or the queue is kept
// full (a deeper ring), and (d) whether the source window is L2-resident or streams from HBM.  The bf16 GEMM measures 6.5 TB/s of
// tile traffic chip-wide (25 GB/s per CU) at 840 TF -- is that the transport's ceiling or the k-loop's drain pattern?
//
// extern "C" double ldsdma_fill(int waves, int depth, int mode, int window_kb, int blocks, int iters, double* ms)
//   mode 0: [issue `depth` pieces per wave, s_waitcnt vmcnt(0), s_barrier] x iters          (drain per batch, like the GEMM)
//   mode 1: keep `depth`..2*depth pieces in flight per wave: issue a batch, wait for the PREVIOUS batch only, no barrier
//   returns TB/s over the whole launch; 128 KB of dynamic LDS per block -> one block per CU
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DMA_PIECE(GSRC, LDSB)                                                                                 \
  {                                                                                                           \
    unsigned keep_;                                                                                           \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(GSRC), "s"(LDSB) : "memory");                                           \
  }

// sharing of the source window (r02): 0 = private window per block; 1 = the blocks of one XCD (blockIdx % 8) share one window and walk it in
// lockstep (what the GEMM's weight operand looks like: every CU asks its L2 for the same lines at the same time -- do those requests
// merge in the L2, or does each become a fabric read?); 2 = same, but block j of the XCD starts j batches into the window (staggered).
__device__ int g_share = 0;
static int h_share = 0;
extern "C" void ldsdma_fill_set_share(int s) { h_share = s; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_share), &s, sizeof(int)); }

template <int DEPTH, int MODE>
__global__ __launch_bounds__(512) void fill_kernel(const char* __restrict__ src, size_t window, int npos, int iters, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwave = blockDim.x >> 6;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // every wave owns 2 * DEPTH ring slots of 1 KiB in the LDS
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * (2u * DEPTH * 1024u));
  const int share = g_share;
  const char* base = src + (size_t)(share ? (blockIdx.x & 7) : blockIdx.x) * window;   // private, or one window per XCD
  const size_t stride = (size_t)nwave * DEPTH * 1024;                     // bytes the block consumes per iteration
  const size_t off0 = (size_t)wave * DEPTH * 1024;                        // wave-uniform; the lane adds 16 bytes of its piece
  int pos = share == 2 ? (int)((blockIdx.x >> 3) * (unsigned)(npos / 32 > 0 ? npos / 32 : 1)) % npos : 0;   // npos = window / stride batches fit in the window
  size_t off = off0 + (size_t)pos * stride;
  const char* lbase = base + (size_t)lane * 16;
  if (MODE == 1) {
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) DMA_PIECE(lbase + off + (size_t)j * 1024, lds_wave + (unsigned)(j * 1024))
    ++pos; off += stride; if (pos == npos) { pos = 0; off = off0; }
  }
  for (int it = 0; it < iters; ++it) {
    const unsigned half = (MODE == 1) ? (unsigned)(((it + 1) & 1) * DEPTH * 1024) : 0u;
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) DMA_PIECE(lbase + off + (size_t)j * 1024, lds_wave + half + (unsigned)(j * 1024))
    ++pos; off += stride; if (pos == npos) { pos = 0; off = off0; }
    if (MODE == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    } else {
      // DEPTH newer pieces may stay in flight: the previous batch has landed
      if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (float)smem[(blockIdx.x * 37) & 1023];
}

template <int DEPTH, int MODE>
static float run(int waves, size_t window, int blocks, int iters, const char* src, float* out) {
  auto kern = fill_kernel<DEPTH, MODE>;
  const int lds = 128 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int npos = (int)(window / ((size_t)waves * DEPTH * 1024));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * waves), lds, 0, src, window, npos, iters / 4 + 1, out);      // warm-up (also warms L2)
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * waves), lds, 0, src, window, npos, iters, out);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

extern "C" double ldsdma_fill(int waves, int depth, int mode, int window_kb, int blocks, int iters, double* ms_out) {
  if (waves < 1 || waves > 8 || (waves * 2 * depth) > 128) return -1.0;
  const size_t window = (size_t)window_kb * 1024;
  if (window < (size_t)waves * depth * 1024 * 2) return -1.0;
  char* src; float* out;
  if (hipMalloc((void**)&src, window * blocks) != hipSuccess) return -2.0;
  (void)hipMemset(src, 1, window * blocks);
  (void)hipMalloc((void**)&out, blocks * sizeof(float));
  float ms = -1.f;
#define CASE(D)                                                                                              \
  if (depth == D) ms = mode == 0 ? run<D, 0>(waves, window, blocks, iters, src, out) : run<D, 1>(waves, window, blocks, iters, src, out);
  CASE(1) CASE(2) CASE(4) CASE(8)
  (void)hipFree(src); (void)hipFree(out);
  if (ms < 0.f) return -3.0;
  *ms_out = ms;
  const double bytes = (double)blocks * waves * depth * 1024.0 * ((double)iters + (mode == 1 ? 1.0 : 0.0));
  return bytes / (ms * 1e-3) / 1e12;
}
