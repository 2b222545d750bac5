// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of librapflow.
// wave = 64 lanes everywhere; no other architecture is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define RAP_OK 0
// rap_set_tuning knobs: process-global A/B switches read on every launch; atomic so that a concurrent rap_set_tuning is a race on
// the VALUE a launch sees, not undefined behaviour.  Host code only.
#include <atomic>
typedef std::atomic<int> rap_tuning_t;

#define RAP_ERR_INVALID (-1)   // bad argument (shape / null / unsupported size)
#define RAP_ERR_WORKSPACE (-2) // workspace too small
#define RAP_ERR_HIP (-3)       // HIP runtime error (hipGetLastError recorded)
#define RAP_ERR_ALLOC (-4)

#define RAP_HIP_CHECK(expr)                                  \
  do {                                                       \
    hipError_t _e = (expr);                                  \
    if (_e != hipSuccess) { rap_set_last_hip_error((int)_e); return RAP_ERR_HIP; } \
  } while (0)

#define RAP_LAUNCH_CHECK()                                   \
  do {                                                       \
    hipError_t _e = hipGetLastError();                       \
    if (_e != hipSuccess) { rap_set_last_hip_error((int)_e); return RAP_ERR_HIP; } \
  } while (0)

void rap_set_last_hip_error(int e);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Row index (within a 32x32 MFMA C/D tile) held in accumulator register `reg` by a lane whose
// half-wave index is `hi` (= lane >> 5).  The column is lane & 31.  (CDNA4 32x32 C/D layout.)
__device__ __forceinline__ int mfma32_crow(int reg, int hi) { return (reg & 3) + 8 * (reg >> 2) + 4 * hi; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// XCD-aware, bijective remap of a 1-D block id: the dispatcher places block b on XCD b % 8, so give
// every XCD one contiguous chunk of the logical tile space (neighbouring tiles share operand panels
// in that XCD's private L2).  Pure speed choice; correct for any placement.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Round a product to fp32 and keep it from being contracted into an FMA with a following add/sub
// (hipcc's default -ffp-contract=fast fuses across statements, and HIP's __fmul_rn is a plain `*`).
__device__ __forceinline__ float mul_rn_nofuse(float a, float b) {
  float m = a * b;
  asm volatile("" : "+v"(m));
  return m;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
