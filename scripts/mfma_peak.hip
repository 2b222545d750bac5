// Microbenchmark: what does v_mfma_f32_32x32x2_f32 sustain on this MI355X with RANDOM operands (the DVFS-limited
// ceiling the fp32 kernels can actually reach), vs zero operands (the datasheet-clock ceiling)?
// extern "C" double mfma_peak(int blocks, int iters, int random, double* eff_clock_ghz)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters,
                                                 unsigned long long* __restrict__ cycles) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  float a0 = in[gid * 4 + 0], a1 = in[gid * 4 + 1], b0 = in[gid * 4 + 2], b1 = in[gid * 4 + 3];
  f32x16 c0, c1, c2, c3;
  for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c3, 0, 0, 0);
    }
    // keep the accumulators bounded without changing the instruction mix much (one VALU per 32 MFMAs)
    a0 = -a0; 
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[gid] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

extern "C" double mfma_peak(int blocks, int iters, int random, double* cycles_per_mfma, double* wall_ms) {
  const size_t n = (size_t)blocks * 256;
  float *in, *out; unsigned long long* cyc;
  hipMalloc(&in, n * 4 * sizeof(float)); hipMalloc(&out, n * sizeof(float)); hipMalloc(&cyc, blocks * sizeof(unsigned long long));
  float* h = (float*)malloc(n * 4 * sizeof(float));
  uint32_t st = 12345u;
  for (size_t i = 0; i < n * 4; ++i) { st = st * 1664525u + 1013904223u; h[i] = random ? ((st >> 8) / 8388608.0f - 1.0f) : 0.0f; }
  hipMemcpy(in, h, n * 4 * sizeof(float), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, in, out, iters / 4 + 1, cyc);   // warm-up
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, in, out, iters, cyc);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long* hc = (unsigned long long*)malloc(blocks * sizeof(unsigned long long));
  hipMemcpy(hc, cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double csum = 0; for (int i = 0; i < blocks; ++i) csum += (double)hc[i];
  const double mfma_per_wave = (double)iters * 32.0;
  *cycles_per_mfma = csum / blocks / mfma_per_wave;       // s_memtime ticks (constant 100 MHz on gfx950) per MFMA per wave
  *wall_ms = ms;
  const double flops = (double)blocks * 4.0 * mfma_per_wave * 2.0 * 32 * 32 * 2;
  hipFree(in); hipFree(out); hipFree(cyc); free(h); free(hc);
  return flops / (ms * 1e-3) / 1e12;
}
