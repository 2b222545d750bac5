// Ablation ladder for the fp32 MFMA GEMM main loop on MI355X: start from a pure v_mfma_f32_32x32x2_f32 stream and add,
// one at a time, what the real kernel does between MFMAs.  Two 256-thread blocks per CU (LDS sized to force that).
//   mode 0: 64 MFMAs per iteration, operands in registers
//   mode 1: + operands re-read from LDS (4 x ds_read_b128 per 16 MFMAs, prefetched one group ahead)
//   mode 2: + one __syncthreads() per 64 MFMAs
//   mode 3: + 8 global_load_dwordx4 per iteration, parked in LDS with 8 ds_write_b128 (the full GEMM loop body)
//   mode 4: mode 3 without the barrier
// extern "C" double ablate(int mode, int blocks, int iters, double* ms)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LD 36

template <int MODE>
__global__ __launch_bounds__(256, 2) void loop_kernel(const float* __restrict__ gsrc, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * 128 * LD];   // 73,728 B -> two blocks per CU
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  for (int i = tid; i < 2 * 2 * 128 * LD; i += 256) smem[i] = gsrc[(blockIdx.x * 977 + i) & 0xfffff];
  __syncthreads();
  const int a_off = ((wave >> 1) * 64 + l31) * LD + 4 * hi;
  const int b_off = 2 * 128 * LD + ((wave & 1) * 64 + l31) * LD + 4 * hi;
  const int srow = tid >> 3, sc4 = (tid & 7) * 4;
  const float* gp = gsrc + ((size_t)blockIdx.x * 128 + srow) * 512 + sc4;
  f32x16 c00, c01, c10, c11;
  for (int r = 0; r < 16; ++r) { c00[r] = 0.f; c01[r] = 0.f; c10[r] = 0.f; c11[r] = 0.f; }
  float4 a0 = *reinterpret_cast<const float4*>(&smem[a_off]);
  float4 a1 = *reinterpret_cast<const float4*>(&smem[a_off + 32 * LD]);
  float4 b0 = *reinterpret_cast<const float4*>(&smem[b_off]);
  float4 b1 = *reinterpret_cast<const float4*>(&smem[b_off + 32 * LD]);
  float4 g0, g1, g2, g3, g4, g5, g6, g7;
#define MF(A0, A1, B0, B1)                                                       \
  c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B0, c00, 0, 0, 0);              \
  c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B1, c01, 0, 0, 0);              \
  c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B0, c10, 0, 0, 0);              \
  c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B1, c11, 0, 0, 0);
#define GROUP(A0, A1, B0, B1) MF(A0.x, A1.x, B0.x, B1.x) MF(A0.y, A1.y, B0.y, B1.y) MF(A0.z, A1.z, B0.z, B1.z) MF(A0.w, A1.w, B0.w, B1.w)
#define FENCE __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < iters; ++it) {
    const int buf = (it & 1) * (128 * LD);
    if (MODE >= 3) {
      const float* q = gp + (size_t)((it * 32) & 511);
      g0 = *reinterpret_cast<const float4*>(q);            g1 = *reinterpret_cast<const float4*>(q + 32 * 512);
      g2 = *reinterpret_cast<const float4*>(q + 64 * 512); g3 = *reinterpret_cast<const float4*>(q + 96 * 512);
      g4 = *reinterpret_cast<const float4*>(q + 8);        g5 = *reinterpret_cast<const float4*>(q + 32 * 512 + 8);
      g6 = *reinterpret_cast<const float4*>(q + 64 * 512 + 8); g7 = *reinterpret_cast<const float4*>(q + 96 * 512 + 8);
      FENCE
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
      if (MODE >= 1) {
        const int o = 8 * ((g + 1) & 3);
        na0 = *reinterpret_cast<const float4*>(&smem[buf + a_off + o]);
        na1 = *reinterpret_cast<const float4*>(&smem[buf + a_off + 32 * LD + o]);
        nb0 = *reinterpret_cast<const float4*>(&smem[buf + b_off + o]);
        nb1 = *reinterpret_cast<const float4*>(&smem[buf + b_off + 32 * LD + o]);
        FENCE
      }
      if (MODE >= 3 && g == 2) {
        float* w = &smem[((it + 1) & 1) * (128 * LD) + srow * LD + sc4];
        *reinterpret_cast<float4*>(w) = g0;                    *reinterpret_cast<float4*>(w + 32 * LD) = g1;
        *reinterpret_cast<float4*>(w + 64 * LD) = g2;          *reinterpret_cast<float4*>(w + 96 * LD) = g3;
        *reinterpret_cast<float4*>(w + 2 * 128 * LD) = g4;     *reinterpret_cast<float4*>(w + 2 * 128 * LD + 32 * LD) = g5;
        *reinterpret_cast<float4*>(w + 2 * 128 * LD + 64 * LD) = g6; *reinterpret_cast<float4*>(w + 2 * 128 * LD + 96 * LD) = g7;
      }
      if ((MODE == 2 || MODE == 3) && g == 3) __syncthreads();
      GROUP(a0, a1, b0, b1)
      if (MODE >= 1) { FENCE }
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
      if (MODE == 0) { a0.x = -a0.x; }
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c00[r] + c01[r] + c10[r] + c11[r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
static float run(int blocks, int iters, const float* src, float* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(loop_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, src, out, iters / 8 + 1);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(loop_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

extern "C" double ablate(int mode, int blocks, int iters, double* ms_out) {
  float *src, *out;
  const size_t nsrc = (size_t)(blocks * 128 + 128) * 512 + (1 << 20);
  (void)hipMalloc(&src, nsrc * sizeof(float)); (void)hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
  float* h = (float*)malloc(nsrc * sizeof(float));
  uint32_t st = 777u;
  for (size_t i = 0; i < nsrc; ++i) { st = st * 1664525u + 1013904223u; h[i] = ((st >> 8) / 8388608.0f - 1.0f) * 1e-3f; }
  (void)hipMemcpy(src, h, nsrc * sizeof(float), hipMemcpyHostToDevice);
  float ms = 0.f;
  switch (mode) {
    case 0: ms = run<0>(blocks, iters, src, out); break;
    case 1: ms = run<1>(blocks, iters, src, out); break;
    case 2: ms = run<2>(blocks, iters, src, out); break;
    case 3: ms = run<3>(blocks, iters, src, out); break;
    default: ms = run<4>(blocks, iters, src, out); break;
  }
  (void)hipFree(src); (void)hipFree(out); free(h);
  *ms_out = ms;
  return (double)blocks * 4.0 * iters * 64.0 * 4096.0 / (ms * 1e-3) / 1e12;
}
