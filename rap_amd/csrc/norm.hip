// HBM-bound normalisation kernels of the velocity network (gfx950, wave = 64).
//
//  * layernorm_kernel: LayerNorm(eps 1e-5, no affine) followed by either the adaLN modulation
//    x * (1 + scale_b) + shift_b  (reference flow_model/norm.py:74-76) or the affine gain/bias of
//    nn.LayerNorm (flow_model/layer.py:88,163).  One wave per token row, float4 loads, the row
//    stays in registers between the statistics pass and the normalisation (4 KiB of traffic per
//    512-wide token: 2 KiB read + 2 KiB write -- the algorithmic minimum for an unfused LN).
//  * qknorm_kernel: MultiHeadRMSNorm of q and k (flow_model/norm.py:28-33, layer.py:91-96):
//    x / max(||x||_2, 1e-12) * gamma[h] * sqrt(64), in place on the head-major q and k planes.
#include "kernels.h"

template <int NV>  // NV float4 per lane: d = 256 * NV
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, float* __restrict__ out, int TP,
                                                        const float* __restrict__ gain_base, const float* __restrict__ shift_base,
                                                        long row_stride, const int32_t* __restrict__ token_row, int add_one) {
  const int d = 256 * NV;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= TP) return;
  const float* xr = x + (size_t)row * d;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = *reinterpret_cast<const float4*>(xr + (i * 64 + lane) * 4);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
    q += (a * a + b * b) + (c * c + e * e);
  }
  const float var = wave_sum(q) / (float)d;
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  const long mrow = token_row ? (long)token_row[row] : 0;
  const float* g = gain_base + mrow * row_stride;
  const float* b = shift_base + mrow * row_stride;
  const float one = add_one ? 1.0f : 0.0f;
  float* orow = out + (size_t)row * d;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    const float4 gg = *reinterpret_cast<const float4*>(g + c);
    const float4 bb = *reinterpret_cast<const float4*>(b + c);
    float4 o;
    o.x = (v[i].x - mean) * rstd * (one + gg.x) + bb.x;
    o.y = (v[i].y - mean) * rstd * (one + gg.y) + bb.y;
    o.z = (v[i].z - mean) * rstd * (one + gg.z) + bb.z;
    o.w = (v[i].w - mean) * rstd * (one + gg.w) + bb.w;
    *reinterpret_cast<float4*>(orow + c) = o;
  }
}

static int launch_ln(hipStream_t stream, const float* x, float* out, int TP, int d, const float* gain, const float* shift,
                     long row_stride, const int32_t* token_row, int add_one) {
  if (TP <= 0) return RAP_OK;
  if (d % 256 != 0 || d > 1024) return RAP_ERR_INVALID;
  dim3 grid((TP + 3) / 4), block(256);
  switch (d / 256) {
    case 1: hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, stream, x, out, TP, gain, shift, row_stride, token_row, add_one); break;
    case 2: hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, stream, x, out, TP, gain, shift, row_stride, token_row, add_one); break;
    case 3: hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, stream, x, out, TP, gain, shift, row_stride, token_row, add_one); break;
    default: hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, stream, x, out, TP, gain, shift, row_stride, token_row, add_one); break;
  }
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

int launch_layernorm_mod(hipStream_t stream, const float* x, float* out, int TP, int d, const float* mod, long mod_stride,
                         const int32_t* token_row) {
  // mod row = [scale (d) | shift (d)]   (norm.py:73: chunk -> scale first, shift second)
  return launch_ln(stream, x, out, TP, d, mod, mod + d, mod_stride, token_row, 1);
}

int launch_layernorm_affine(hipStream_t stream, const float* x, float* out, int TP, int d, const float* gain,
                            const float* shift) {
  return launch_ln(stream, x, out, TP, d, gain, shift, 0, nullptr, 0);
}

// 16 lanes per (plane, head, token) row of 64 floats; 16 rows per 256-thread block.
__global__ __launch_bounds__(256) void qknorm_kernel(float* __restrict__ qk, long rows_per_plane, int TP, int heads,
                                                     const float* __restrict__ gamma_q, const float* __restrict__ gamma_k) {
  const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (row >= 2 * rows_per_plane) return;
  const int sub = threadIdx.x & 15;
  const int plane = row >= rows_per_plane ? 1 : 0;
  const long r = row - (long)plane * rows_per_plane;
  const int head = (int)(r / TP);
  float* p = qk + row * 64 + sub * 4;
  float4 v = *reinterpret_cast<const float4*>(p);
  float s = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float nrm = fmaxf(sqrtf(s), 1e-12f);
  const float4 g = *reinterpret_cast<const float4*>((plane ? gamma_k : gamma_q) + head * 64 + sub * 4);
  v.x = v.x / nrm * g.x * 8.0f;
  v.y = v.y / nrm * g.y * 8.0f;
  v.z = v.z / nrm * g.z * 8.0f;
  v.w = v.w / nrm * g.w * 8.0f;
  *reinterpret_cast<float4*>(p) = v;
}

int launch_qknorm(hipStream_t stream, float* qkv_headmajor, int TP, int heads, const float* gamma_q, const float* gamma_k) {
  if (TP <= 0) return RAP_OK;
  const long rows_per_plane = (long)heads * TP;
  const long nblk = (2 * rows_per_plane + 15) / 16;
  hipLaunchKernelGGL(qknorm_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, qkv_headmajor, rows_per_plane, TP, heads,
                     gamma_q, gamma_k);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
