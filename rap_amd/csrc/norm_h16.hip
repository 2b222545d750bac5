// HBM-bound normalisation kernels of the reduced-precision path: same arithmetic as norm.hip (fp32 statistics,
// fp32 modulation), 16-bit outputs for the bf16 / fp16 GEMM and attention operands.
//  * layernorm_h16_kernel: fp32 residual stream in, LN (+ adaLN modulation or affine), 16-bit out:
//    3 KiB of traffic per 512-wide token (2 KiB read + 1 KiB write); XH = true: the residual stream is fp16 (round 3,
//    rap_model_set_residual_dtype): 2 KiB per token.
//  * qknorm_h16_kernel: MultiHeadRMSNorm (flow_model/norm.py:28-33) in place on the 16-bit q and k planes.
#include "half.h"
#include "kernels.h"

// COMB (round 6, few-token calls): the kernel is ALSO the combine pass of the residual GEMM in front of it.  The GEMM leaves `splits` fp32
// partial planes (no bias, no residual); a row's new residual-stream value is formed here exactly as gemm_h16_splitk_combine_kernel forms
// it -- partial planes in order, then the bias, then the residual; ONE rounding (saturating) when the stream is fp16 -- stored, and
// normalised from the STORED value: bit-identical to the combine pass followed by the plain LayerNorm, one launch and one round trip of
// the row through HBM / L2 fewer per LayerNorm (three per layer).
struct LnCombine {
  const float* part;      // [splits][rows][d] fp32
  int splits;
  long plane_stride;      // rows * d
  const float* bias;      // (d) or null
  void* h_out;            // the residual stream (fp32, or fp16 when XH); may alias x
};
template <int NV, int DT, bool XH, bool COMB = false>  // NV float4 per lane: d = 256 * NV
__global__ __launch_bounds__(256) void layernorm_h16_kernel(const void* __restrict__ x_, u16* __restrict__ out, int TP,
                                                            const float* __restrict__ gain_base, const float* __restrict__ shift_base,
                                                            long row_stride, const int32_t* __restrict__ token_row, int add_one, LnCombine cb) {
  const int d = 256 * NV;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= TP) return;
  float4 v[NV];
  float s = 0.f;
  [[maybe_unused]] float4 pacc[NV];
  if constexpr (COMB) {
    constexpr bool WIDE_ = XH && NV % 2 == 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = WIDE_ ? (i / 2) * 512 + lane * 8 + (i % 2) * 4 : (i * 64 + lane) * 4;
      float4 a = float4{0.f, 0.f, 0.f, 0.f};
      for (int sp = 0; sp < cb.splits; ++sp) {
        const float4 t = *reinterpret_cast<const float4*>(cb.part + (size_t)sp * cb.plane_stride + (size_t)row * d + c);
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
      }
      if (cb.bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(cb.bias + c);
        a.x += b4.x; a.y += b4.y; a.z += b4.z; a.w += b4.w;
      }
      pacc[i] = a;
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if constexpr (XH && NV % 2 == 0) {
      // 16 bytes per lane in, 16 bytes per lane out: v[i], v[i + 1] are 8 CONSECUTIVE columns (col_of below)
      if (i % 2 == 0) {
        float f8[8];
        h16_unpack8<RAP_DT_F16>(*reinterpret_cast<const uint4*>(reinterpret_cast<const u16*>(x_) + (size_t)row * d + (i / 2) * 512 + lane * 8), f8);
        v[i] = float4{f8[0], f8[1], f8[2], f8[3]};
        v[i + 1] = float4{f8[4], f8[5], f8[6], f8[7]};
      }
    } else if constexpr (XH) {
      const uint2 raw = *reinterpret_cast<const uint2*>(reinterpret_cast<const u16*>(x_) + (size_t)row * d + (i * 64 + lane) * 4);
      const typename H16<RAP_DT_F16>::T4 h4 = __builtin_bit_cast(typename H16<RAP_DT_F16>::T4, raw);
      v[i] = float4{(float)h4[0], (float)h4[1], (float)h4[2], (float)h4[3]};
    } else {
      v[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(x_) + (size_t)row * d + (i * 64 + lane) * 4);
    }
  }
  if constexpr (COMB) {
    constexpr bool WIDE_ = XH && NV % 2 == 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i].x += pacc[i].x; v[i].y += pacc[i].y; v[i].z += pacc[i].z; v[i].w += pacc[i].w; }     // (partials + bias) + residual
    if constexpr (XH) {
      u16* hrow = reinterpret_cast<u16*>(cb.h_out) + (size_t)row * d;
      if constexpr (WIDE_) {
#pragma unroll
        for (int i = 0; i < NV; i += 2) {
          const typename H16<RAP_DT_F16>::T8 o8 = f16_pack8_sat(v[i].x, v[i].y, v[i].z, v[i].w, v[i + 1].x, v[i + 1].y, v[i + 1].z, v[i + 1].w);
          *reinterpret_cast<uint4*>(hrow + (i / 2) * 512 + lane * 8) = __builtin_bit_cast(uint4, o8);
          float f8[8];
          h16_unpack8<RAP_DT_F16>(__builtin_bit_cast(uint4, o8), f8);             // the LayerNorm sees the STORED (rounded) stream value
          v[i] = float4{f8[0], f8[1], f8[2], f8[3]};
          v[i + 1] = float4{f8[4], f8[5], f8[6], f8[7]};
        }
      } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const typename H16<RAP_DT_F16>::T8 o8 = f16_pack8_sat(v[i].x, v[i].y, v[i].z, v[i].w, 0.f, 0.f, 0.f, 0.f);
          const uint4 raw = __builtin_bit_cast(uint4, o8);
          *reinterpret_cast<uint2*>(hrow + (i * 64 + lane) * 4) = make_uint2(raw.x, raw.y);
          float f8[8];
          h16_unpack8<RAP_DT_F16>(raw, f8);
          v[i] = float4{f8[0], f8[1], f8[2], f8[3]};
        }
      }
    } else {
      float* hrow = reinterpret_cast<float*>(cb.h_out) + (size_t)row * d;
#pragma unroll
      for (int i = 0; i < NV; ++i) *reinterpret_cast<float4*>(hrow + (i * 64 + lane) * 4) = v[i];
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
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
  u16* orow = out + (size_t)row * d;
  constexpr bool WIDE = XH && NV % 2 == 0;
  uint2 keep = make_uint2(0, 0);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = WIDE ? (i / 2) * 512 + lane * 8 + (i % 2) * 4 : (i * 64 + lane) * 4;
    const float4 gg = *reinterpret_cast<const float4*>(g + c);
    const float4 bb = *reinterpret_cast<const float4*>(b + c);
    const float ox = (v[i].x - mean) * rstd * (one + gg.x) + bb.x;
    const float oy = (v[i].y - mean) * rstd * (one + gg.y) + bb.y;
    const float oz = (v[i].z - mean) * rstd * (one + gg.z) + bb.z;
    const float ow = (v[i].w - mean) * rstd * (one + gg.w) + bb.w;
    const uint2 pk = h16_pack4<DT>(ox, oy, oz, ow);
    if constexpr (WIDE) {
      if (i % 2 == 0) keep = pk;
      else *reinterpret_cast<uint4*>(orow + c - 4) = make_uint4(keep.x, keep.y, pk.x, pk.y);
    } else {
      *reinterpret_cast<uint2*>(orow + c) = pk;
    }
  }
}

// Split-precision twin (RAP_DT_F32X2, round 5): fp32 residual stream in, the LayerNorm output as fp16 head / tail planes in the paired
// layout (half.h: logical column k -> physical x2_col(k), tail 32 further; row stride 2 d).  A lane's 4 consecutive columns lie in one
// 32-column chunk, so it writes 8 bytes of heads and 8 bytes of tails; 8 lanes cover a chunk's 128-byte line.  6 KiB per token.
template <int NV, bool COMB = false>
__global__ __launch_bounds__(256) void layernorm_x2_kernel(const float* __restrict__ x, u16* __restrict__ out, int TP,
                                                           const float* __restrict__ gain_base, const float* __restrict__ shift_base,
                                                           long row_stride, const int32_t* __restrict__ token_row, int add_one, LnCombine cb) {
  const int d = 256 * NV;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= TP) return;
  float4 v[NV];
  float s = 0.f;
  [[maybe_unused]] float4 pacc[NV];
  if constexpr (COMB) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      float4 a = float4{0.f, 0.f, 0.f, 0.f};
      for (int sp = 0; sp < cb.splits; ++sp) {
        const float4 t = *reinterpret_cast<const float4*>(cb.part + (size_t)sp * cb.plane_stride + (size_t)row * d + c);
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
      }
      if (cb.bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(cb.bias + c);
        a.x += b4.x; a.y += b4.y; a.z += b4.z; a.w += b4.w;
      }
      pacc[i] = a;
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(x + (size_t)row * d + (i * 64 + lane) * 4);
  if constexpr (COMB) {
    float* hrow = reinterpret_cast<float*>(cb.h_out) + (size_t)row * d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i].x += pacc[i].x; v[i].y += pacc[i].y; v[i].z += pacc[i].z; v[i].w += pacc[i].w;
      *reinterpret_cast<float4*>(hrow + (i * 64 + lane) * 4) = v[i];
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
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
  u16* orow = out + (size_t)row * (2 * d);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    const float4 gg = *reinterpret_cast<const float4*>(g + c);
    const float4 bb = *reinterpret_cast<const float4*>(b + c);
    const float ox = (v[i].x - mean) * rstd * (one + gg.x) + bb.x;
    const float oy = (v[i].y - mean) * rstd * (one + gg.y) + bb.y;
    const float oz = (v[i].z - mean) * rstd * (one + gg.z) + bb.z;
    const float ow = (v[i].w - mean) * rstd * (one + gg.w) + bb.w;
    uint2 hi, lo;
    x2_split4(ox, oy, oz, ow, hi, lo);
    *reinterpret_cast<uint2*>(orow + x2_col(c)) = hi;
    *reinterpret_cast<uint2*>(orow + x2_col(c) + 32) = lo;
  }
}
template <bool COMB>
static int launch_ln_x2(hipStream_t stream, const void* x, int x_f16, u16* out, int TP, int d, const float* gain, const float* shift,
                        long row_stride, const int32_t* token_row, int add_one, const LnCombine& cb) {
  if (TP <= 0) return RAP_OK;
  if (d % 256 != 0 || d > 1024 || x_f16) return RAP_ERR_INVALID;      // the split mode keeps the residual stream in fp32
  const float* xf = reinterpret_cast<const float*>(x);
  dim3 grid((TP + 3) / 4), block(256);
  switch (d / 256) {
    case 1: hipLaunchKernelGGL((layernorm_x2_kernel<1, COMB>), grid, block, 0, stream, xf, out, TP, gain, shift, row_stride, token_row, add_one, cb); break;
    case 2: hipLaunchKernelGGL((layernorm_x2_kernel<2, COMB>), grid, block, 0, stream, xf, out, TP, gain, shift, row_stride, token_row, add_one, cb); break;
    case 3: hipLaunchKernelGGL((layernorm_x2_kernel<3, COMB>), grid, block, 0, stream, xf, out, TP, gain, shift, row_stride, token_row, add_one, cb); break;
    default: hipLaunchKernelGGL((layernorm_x2_kernel<4, COMB>), grid, block, 0, stream, xf, out, TP, gain, shift, row_stride, token_row, add_one, cb); break;
  }
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

template <int DT, bool XH, bool COMB>
static int launch_ln_h16(hipStream_t stream, const void* x, u16* out, int TP, int d, const float* gain, const float* shift,
                         long row_stride, const int32_t* token_row, int add_one, const LnCombine& cb) {
  if (TP <= 0) return RAP_OK;
  if (d % 256 != 0 || d > 1024) return RAP_ERR_INVALID;
  dim3 grid((TP + 3) / 4), block(256);
  switch (d / 256) {
    case 1: hipLaunchKernelGGL((layernorm_h16_kernel<1, DT, XH, COMB>), grid, block, 0, stream, x, out, TP, gain, shift, row_stride, token_row, add_one, cb); break;
    case 2: hipLaunchKernelGGL((layernorm_h16_kernel<2, DT, XH, COMB>), grid, block, 0, stream, x, out, TP, gain, shift, row_stride, token_row, add_one, cb); break;
    case 3: hipLaunchKernelGGL((layernorm_h16_kernel<3, DT, XH, COMB>), grid, block, 0, stream, x, out, TP, gain, shift, row_stride, token_row, add_one, cb); break;
    default: hipLaunchKernelGGL((layernorm_h16_kernel<4, DT, XH, COMB>), grid, block, 0, stream, x, out, TP, gain, shift, row_stride, token_row, add_one, cb); break;
  }
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
template <bool COMB>
static int launch_ln_h16_any(hipStream_t stream, int dtype, const void* x, int x_f16, u16* out, int TP, int d, const float* gain,
                             const float* shift, long row_stride, const int32_t* token_row, int add_one, const LnCombine& cb) {
  if (dtype == RAP_DT_F32X2) return launch_ln_x2<COMB>(stream, x, x_f16, out, TP, d, gain, shift, row_stride, token_row, add_one, cb);
  if (dtype == RAP_DT_BF16)
    return x_f16 ? launch_ln_h16<RAP_DT_BF16, true, COMB>(stream, x, out, TP, d, gain, shift, row_stride, token_row, add_one, cb)
                 : launch_ln_h16<RAP_DT_BF16, false, COMB>(stream, x, out, TP, d, gain, shift, row_stride, token_row, add_one, cb);
  if (dtype == RAP_DT_F16)
    return x_f16 ? launch_ln_h16<RAP_DT_F16, true, COMB>(stream, x, out, TP, d, gain, shift, row_stride, token_row, add_one, cb)
                 : launch_ln_h16<RAP_DT_F16, false, COMB>(stream, x, out, TP, d, gain, shift, row_stride, token_row, add_one, cb);
  return RAP_ERR_INVALID;
}

int launch_layernorm_mod_h16(hipStream_t stream, int dtype, const void* x, int x_f16, u16* out, int TP, int d, const float* mod,
                             long mod_stride, const int32_t* token_row) {
  return launch_ln_h16_any<false>(stream, dtype, x, x_f16, out, TP, d, mod, mod + d, mod_stride, token_row, 1, LnCombine{});
}
int launch_layernorm_affine_h16(hipStream_t stream, int dtype, const void* x, int x_f16, u16* out, int TP, int d, const float* gain,
                                const float* shift) {
  return launch_ln_h16_any<false>(stream, dtype, x, x_f16, out, TP, d, gain, shift, 0, nullptr, 0, LnCombine{});
}
// combine pass of a split-K residual GEMM + the LayerNorm that follows it (see LnCombine): h (fp32 / fp16, in place) = h + bias + sum of
// the `splits` partial planes part[s][rows][d]; out = LN(h) modulated (mod != null: 1 + mod[0:d], mod[d:2d]) or affine (gain, shift)
int launch_resid_combine_ln_h16(hipStream_t stream, int dtype, const float* part, int splits, const float* bias, void* h, int h_f16, u16* out,
                                int rows, int d, const float* mod, long mod_stride, const int32_t* token_row, const float* gain,
                                const float* shift) {
  if (!part || splits < 1 || splits > 8 || !h || !out) return RAP_ERR_INVALID;
  const LnCombine cb{part, splits, (long)rows * d, bias, h};
  if (mod) return launch_ln_h16_any<true>(stream, dtype, h, h_f16, out, rows, d, mod, mod + d, mod_stride, token_row, 1, cb);
  return launch_ln_h16_any<true>(stream, dtype, h, h_f16, out, rows, d, gain, shift, 0, nullptr, 0, cb);
}

// 8 lanes per (plane, head, token) row of 64 values (16 bytes per lane); 32 rows per 256-thread block.
template <int DT>
__global__ __launch_bounds__(256) void qknorm_h16_kernel(u16* __restrict__ qk, long rows_per_plane, int TP, int heads,
                                                         const float* __restrict__ gamma_q, const float* __restrict__ gamma_k,
                                                         float q_mul) {
  const long row = (long)blockIdx.x * 32 + (threadIdx.x >> 3);
  if (row >= 2 * rows_per_plane) return;
  const int sub = threadIdx.x & 7;
  const int plane = row >= rows_per_plane ? 1 : 0;
  const long r = row - (long)plane * rows_per_plane;
  const int head = (int)(r / TP);
  u16* p = qk + row * 64 + sub * 8;
  float v[8];
  h16_unpack8<DT>(*reinterpret_cast<const uint4*>(p), v);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i] * v[i];
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float nrm = fmaxf(sqrtf(s), 1e-12f);
  const float* g = (plane ? gamma_k : gamma_q) + head * 64 + sub * 8;
  const float4 g0 = *reinterpret_cast<const float4*>(g);
  const float4 g1 = *reinterpret_cast<const float4*>(g + 4);
  // the reference's factor sqrt(64) = 8 (norm.py:33); the q plane may instead be written PRE-SCALED for the attention kernel:
  // q_mul = 8 * (log2(e) / 8) folds the softmax scale and the exp -> exp2 conversion into this (single) rounding of q
  const float mul = plane ? 8.0f : q_mul;
  const typename H16<DT>::T8 o8 = h16_pack8<DT>(v[0] / nrm * g0.x * mul, v[1] / nrm * g0.y * mul, v[2] / nrm * g0.z * mul,
                                                v[3] / nrm * g0.w * mul, v[4] / nrm * g1.x * mul, v[5] / nrm * g1.y * mul,
                                                v[6] / nrm * g1.z * mul, v[7] / nrm * g1.w * mul);
  *reinterpret_cast<uint4*>(p) = __builtin_bit_cast(uint4, o8);
}

int launch_qknorm_h16(hipStream_t stream, int dtype, u16* qk, int TP, int heads, const float* gamma_q, const float* gamma_k,
                      float q_mul) {
  if (TP <= 0) return RAP_OK;
  const long rows_per_plane = (long)heads * TP;
  const long nblk = (2 * rows_per_plane + 31) / 32;
  if (dtype == RAP_DT_BF16)
    hipLaunchKernelGGL(qknorm_h16_kernel<RAP_DT_BF16>, dim3((unsigned)nblk), dim3(256), 0, stream, qk, rows_per_plane, TP, heads, gamma_q, gamma_k, q_mul);
  else if (dtype == RAP_DT_F16)
    hipLaunchKernelGGL(qknorm_h16_kernel<RAP_DT_F16>, dim3((unsigned)nblk), dim3(256), 0, stream, qk, rows_per_plane, TP, heads, gamma_q, gamma_k, q_mul);
  else
    return RAP_ERR_INVALID;
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

// Upper bound of the attention logits of one (layer, branch) after MultiHeadRMSNorm: q = x/|x| * gamma_q * 8, so
// |q| <= 8 max|gamma_q| (same for k) and q.k/8 <= 8 max|gamma_q| max|gamma_k|.  One wave per head.
__global__ __launch_bounds__(64) void qk_logit_bound_kernel(const float* __restrict__ gq, const float* __restrict__ gk,
                                                            float* __restrict__ out) {
  const int h = blockIdx.x, lane = threadIdx.x;
  float a = fabsf(gq[h * 64 + lane]), b = fabsf(gk[h * 64 + lane]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a = fmaxf(a, __shfl_xor(a, o, 64)); b = fmaxf(b, __shfl_xor(b, o, 64)); }
  if (lane == 0) out[h] = 8.0f * a * b * 1.001f;      // 0.1 % slack for the fp32 roundings inside qknorm
}
int launch_qk_logit_bound(hipStream_t stream, const float* gamma_q, const float* gamma_k, int heads, float* out) {
  hipLaunchKernelGGL(qk_logit_bound_kernel, dim3(heads), dim3(64), 0, stream, gamma_q, gamma_k, out);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
