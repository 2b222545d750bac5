// 16-bit operand types of the reduced-precision velocity-network path (gfx950 only).
//
// DT = 1: bfloat16 (BASELINE configs[2]/[4] "bf16 MFMA"), DT = 2: IEEE half (what the reference's shipped GPU
// configuration runs: Lightning "16-mixed" autocast + fp16 flash-attn, trainer/infer.yaml:6, layer.py:106-128).
// Both feed v_mfma_f32_32x32x16_{bf16,f16}: 8 elements per lane per operand, fp32 accumulate, 32 cycles per
// instruction per SIMD (16x the rate of the fp32-input MFMA the exact path uses).
//
// Operand layout of the 32x32x16 forms (A is 32 x 16, B is 16 x 32): lane l holds row/column (l & 31) and the
// eight k-slots 8*(l >> 5) .. +7.  The contraction runs over (half-wave, slot) pairs, so ANY assignment of
// logical k indices to (step, half-wave, slot) is valid as long as A and B use the same one -- the attention
// kernel uses that freedom to feed P straight from the accumulator registers.
#pragma once
#include "common.h"

#define RAP_DT_F32 0
#define RAP_DT_BF16 1
#define RAP_DT_F16 2
// Round 5: fp32-ACCURATE arithmetic on the fp16 matrix pipe ("split precision").  Every operand x is carried as an fp16 head and an
// fp16 tail, x = hi + lo (hi = rn16(x), lo = rn16(x - hi): 22 significand bits), and a contraction keeps the three products
// hi*hi + hi*lo + lo*hi in ONE fp32 accumulator (a product of two fp16 values is exact in fp32; the dropped lo*lo term is 2^-22
// relative).  Storage is the "paired" layout: a logical row of K values is 2K fp16 values, chunk c = k >> 5 of 32 logical columns
// occupying physical columns [64c, 64c + 32) = heads and [64c + 32, 64c + 64) = tails -- one 128-byte line per chunk, so the
// LDS-DMA / swizzle / fragment-read machinery of the 16-bit kernels serves it unchanged with K_physical = 2 K.
#define RAP_DT_F32X2 3

typedef unsigned short u16;
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int DT> struct H16;

template <> struct H16<RAP_DT_BF16> {
  typedef __bf16 T;
  typedef __bf16 T2 __attribute__((ext_vector_type(2)));
  typedef __bf16 T4 __attribute__((ext_vector_type(4)));
  typedef __bf16 T8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ f32x16 mfma(T8 a, T8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct H16<RAP_DT_F16> {
  typedef _Float16 T;
  typedef _Float16 T2 __attribute__((ext_vector_type(2)));
  typedef _Float16 T4 __attribute__((ext_vector_type(4)));
  typedef _Float16 T8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ f32x16 mfma(T8 a, T8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// round-to-nearest-even conversions (v_cvt_pk_{bf16,f16}_f32)
template <int DT> __device__ __forceinline__ u16 h16_from_f32(float x) {
  typename H16<DT>::T h = (typename H16<DT>::T)x;
  return __builtin_bit_cast(u16, h);
}
template <int DT> __device__ __forceinline__ float h16_to_f32(u16 x) {
  return (float)__builtin_bit_cast(typename H16<DT>::T, x);
}
template <int DT> __device__ __forceinline__ uint32_t h16_pack2(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, typename H16<DT>::T2));
}
template <int DT> __device__ __forceinline__ uint2 h16_pack4(float a, float b, float c, float d) {
  f32x4 v = {a, b, c, d};
  return __builtin_bit_cast(uint2, __builtin_convertvector(v, typename H16<DT>::T4));
}
template <int DT> __device__ __forceinline__ typename H16<DT>::T8 h16_pack8(float a, float b, float c, float d, float e,
                                                                            float f, float g, float h) {
  f32x8 v = {a, b, c, d, e, f, g, h};
  return __builtin_convertvector(v, typename H16<DT>::T8);
}
// Saturating fp16 pack for the 16-bit RESIDUAL STREAM (round 4): a plain float -> _Float16 conversion turns |x| > 65504
// into inf, which the next LayerNorm turns into NaN for the whole row; the stream instead saturates at the largest finite fp16
// (v_med3_f32 + a NaN select).  Operand and weight conversions do not saturate (RNE, as torch's).
// (v_med3_f32 alone returns the MINIMUM of the other two when one operand is NaN -- it would turn NaN into -65504; ADVICE r04 --
// hence the explicit select: NaN stays NaN as in the reference's own fp16 arithmetic.)
__device__ __forceinline__ float f16_sat(float x) {
  const float m = __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);
  return x != x ? x : m;
}
__device__ __forceinline__ typename H16<RAP_DT_F16>::T8 f16_pack8_sat(float a, float b, float c, float d, float e, float f, float g, float h) {
  return h16_pack8<RAP_DT_F16>(f16_sat(a), f16_sat(b), f16_sat(c), f16_sat(d), f16_sat(e), f16_sat(f), f16_sat(g), f16_sat(h));
}
// ---- split precision (RAP_DT_F32X2): head / tail planes of 8 (or 4) fp32 values.  The head saturates at the largest finite fp16
// (the range of the reference's own fp16 autocast inference); the tail x - hi is exact in fp32 and rounded once.
typedef typename H16<RAP_DT_F16>::T8 x2_t8;
typedef typename H16<RAP_DT_F16>::T4 x2_t4;
__device__ __forceinline__ void x2_split8(const float (&v)[8], x2_t8& hi, x2_t8& lo) {
  f32x8 s;
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = f16_sat(v[i]);          // the value itself is clipped to the fp16 range: head AND tail stay finite
  hi = __builtin_convertvector(s, x2_t8);
  lo = __builtin_convertvector(s - __builtin_convertvector(hi, f32x8), x2_t8);
}
// no saturation: for values bounded by construction (softmax probabilities, attention outputs of finite V).  The tail p - hi is formed by
// v_fma_mix_f32 (fma(hi as fp16, -1.0, p): the difference is exact, one instruction instead of v_cvt_f32_f16 + v_sub_f32 -- hipcc does not
// select it by itself; 32 of the ~250 VALU instructions of an attention key tile)
__device__ __forceinline__ void x2_split8_nosat(const f32x8 v, x2_t8& hi, x2_t8& lo) {
  hi = __builtin_convertvector(v, x2_t8);
  const uint4 hp = __builtin_bit_cast(uint4, hi);
  const unsigned hw[4] = {hp.x, hp.y, hp.z, hp.w};
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hw[i]), "v"(v[2 * i]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hw[i]), "v"(v[2 * i + 1]));
    r[2 * i] = r0; r[2 * i + 1] = r1;
  }
  lo = __builtin_convertvector(r, x2_t8);
}
__device__ __forceinline__ void x2_split4(float a, float b, float c, float d, uint2& hi, uint2& lo) {
  const f32x4 s = {f16_sat(a), f16_sat(b), f16_sat(c), f16_sat(d)};
  const x2_t4 h = __builtin_convertvector(s, x2_t4);
  const x2_t4 l = __builtin_convertvector(s - __builtin_convertvector(h, f32x4), x2_t4);
  hi = __builtin_bit_cast(uint2, h);
  lo = __builtin_bit_cast(uint2, l);
}
// physical column of logical column k in the paired layout (head plane; the tail is 32 further)
__host__ __device__ __forceinline__ int x2_col(int k) { return ((k >> 5) << 6) | (k & 31); }

template <int DT> __device__ __forceinline__ void h16_unpack8(uint4 raw, float (&out)[8]) {
  const typename H16<DT>::T8 v = __builtin_bit_cast(typename H16<DT>::T8, raw);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
}

// Position of token `tok` inside its 64-token block of the transposed V image: bits 2 and 3 of the in-block index
// are swapped, so that the eight keys a lane of the P*V MFMA contracts in one step (accumulator registers
// 8(s&1)..8(s&1)+7 of the S^T tile = keys 16s + 4hi + {0..3} and 16s + 8 + 4hi + {0..3}) are CONTIGUOUS:
// positions 16s + 8hi .. +7 -- one ds_read_b128 per MFMA, no cross-lane traffic between the two products.
__host__ __device__ __forceinline__ int vt_pos(int tok_in_block) {
  return (tok_in_block & ~12) | ((tok_in_block & 4) << 1) | ((tok_in_block & 8) >> 1);
}

// Stage-major over NP independent pairs: a packed result feeds the next packed op only NP instructions later (issued back to back,
// hipcc pads every dependent v_pk pair with an s_nop: 249 of them in the first version).
template <int NP>
__device__ __forceinline__ void geglu_pairs(const f32x2 (&h)[NP], const f32x2 (&g)[NP], f32x2 (&o)[NP]) {
  f32x2 ax[NP], t[NP], poly[NP], q[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) ax[i] = f32x2{fabsf(g[i].x), fabsf(g[i].y)} * f32x2{0.70710678118654752440f, 0.70710678118654752440f};
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const f32x2 d = __builtin_elementwise_fma(ax[i], f32x2{0.3275911f, 0.3275911f}, f32x2{1.0f, 1.0f});
    t[i] = f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const f32x2 e = ax[i] * (ax[i] * f32x2{-1.44269504088896340736f, -1.44269504088896340736f});
    q[i] = f32x2{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) poly[i] = __builtin_elementwise_fma(t[i], f32x2{1.061405429f, 1.061405429f}, f32x2{-1.453152027f, -1.453152027f});
#pragma unroll
  for (int i = 0; i < NP; ++i) poly[i] = __builtin_elementwise_fma(t[i], poly[i], f32x2{1.421413741f, 1.421413741f});
#pragma unroll
  for (int i = 0; i < NP; ++i) poly[i] = __builtin_elementwise_fma(t[i], poly[i], f32x2{-0.284496736f, -0.284496736f});
#pragma unroll
  for (int i = 0; i < NP; ++i) poly[i] = __builtin_elementwise_fma(t[i], poly[i], f32x2{0.254829592f, 0.254829592f});
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] *= poly[i] * t[i];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const f32x2 u = __builtin_elementwise_fma(q[i], f32x2{-0.5f, -0.5f}, f32x2{0.5f, 0.5f});     // 1/2 - q/2 >= 0
    q[i] = f32x2{0.5f, 0.5f} + f32x2{__builtin_copysignf(u.x, g[i].x), __builtin_copysignf(u.y, g[i].y)};
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) o[i] = h[i] * (g[i] * q[i]);
}

