// Split precision (RAP_DT_F32X2, half.h): fp32 matrices -> fp16 head / tail planes in the paired layout, and the per-tensor
// power-of-two scale of the weight planes.
//
// Weights of a trained DiT are small (|w| ~ 1e-2): the tail of such a value is an fp16 SUBNORMAL (quantum 2^-24), i.e. the pair
// would carry an absolute error of 3e-8 = 1.5e-6 relative -- an order of magnitude above fp32 -- and a matrix pipe that flushed
// fp16 subnormals would lose the tails altogether.  Every weight tensor is therefore stored multiplied by 2^e with
// max|w| 2^e in [2^11, 2^12]: tails of all entries within 2^14 of the largest are normal numbers, the products stay far inside fp32,
// and the epilogues multiply the accumulators by 2^-e (exact).  scripts/x2_emulation.py: with the scale the sampled clouds sit
// 2e-7 from fp64 (fp32 oracle: 2.9e-7), and 1.6e-6 even if subnormal operands were flushed (5e-5 without it).
#include "half.h"
#include "kernels.h"

__global__ __launch_bounds__(256) void x2_pack_kernel(const float* __restrict__ src, long ld_src, long rows, int cols, float scale,
                                                      u16* __restrict__ dst) {
  const int c4 = cols / 4;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long n = rows * c4, stride = (long)gridDim.x * 256;
  for (; i < n; i += stride) {
    const long r = i / c4;
    const int c = (int)(i % c4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(src + r * ld_src + c);
    uint2 hi, lo;
    x2_split4(v.x * scale, v.y * scale, v.z * scale, v.w * scale, hi, lo);
    u16* o = dst + r * (2L * cols) + x2_col(c);
    *reinterpret_cast<uint2*>(o) = hi;
    *reinterpret_cast<uint2*>(o + 32) = lo;
  }
}

int launch_x2_pack(hipStream_t stream, const float* src, long ld_src, long rows, int cols, float scale, u16* dst) {
  if (rows <= 0 || cols <= 0) return RAP_OK;
  if (cols % 32 != 0 || (ld_src & 3)) return RAP_ERR_INVALID;
  const long n = rows * (cols / 4);
  const unsigned grid = (unsigned)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
  hipLaunchKernelGGL(x2_pack_kernel, dim3(grid), dim3(256), 0, stream, src, ld_src, rows, cols, scale, dst);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

// paired fp16 -> fp32 (tests, and the inverse of the pack): dst[r][k] = (hi + lo) / scale
__global__ __launch_bounds__(256) void x2_unpack_kernel(const u16* __restrict__ src, long rows, int cols, float inv_scale, float* __restrict__ dst) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long n = rows * cols, stride = (long)gridDim.x * 256;
  for (; i < n; i += stride) {
    const long r = i / cols;
    const int k = (int)(i % cols);
    const u16* p = src + r * (2L * cols) + x2_col(k);
    dst[i] = (h16_to_f32<RAP_DT_F16>(p[0]) + h16_to_f32<RAP_DT_F16>(p[32])) * inv_scale;
  }
}
int launch_x2_unpack(hipStream_t stream, const u16* src, long rows, int cols, float inv_scale, float* dst) {
  if (rows <= 0 || cols <= 0) return RAP_OK;
  if (cols % 32 != 0) return RAP_ERR_INVALID;
  const long n = rows * cols;
  const unsigned grid = (unsigned)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
  hipLaunchKernelGGL(x2_unpack_kernel, dim3(grid), dim3(256), 0, stream, src, rows, cols, inv_scale, dst);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

// max |x| over n floats -> *out (uint bits of a non-negative float order like the float; out must be zeroed by the caller)
__global__ __launch_bounds__(256) void max_abs_kernel(const float* __restrict__ x, size_t n, unsigned* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  float m = 0.f;
  for (; i < n; i += stride) { const float a = fabsf(x[i]); m = a > m ? a : m; }      // NaN compares false: ignored
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}
int launch_max_abs(hipStream_t stream, const float* x, size_t n, float* out) {
  if (n == 0) return RAP_OK;
  const unsigned grid = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(max_abs_kernel, dim3(grid), dim3(256), 0, stream, x, n, reinterpret_cast<unsigned*>(out));
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
