// Variable-length, non-causal softmax attention with fp32-ACCURATE products on the fp16 matrix pipe (split precision, round 5).
//
// Third twin of attn_f32.hip / attn_h16.hip; replaces flash_attn.flash_attn_varlen_qkvpacked_func (flow_model/layer.py:106-111 per
// part, :123-128 per sample) for compute dtype RAP_DT_F32X2: every operand is an fp16 head + an fp16 tail (x = hi + lo, 22
// significand bits, half.h) and both contractions keep hi*hi + hi*lo + lo*hi in the MFMA's fp32 accumulators:
//     S^T = K_hi Q_hi^T + K_hi Q_lo^T + K_lo Q_hi^T        O^T += V_hi^T P_hi^T + V_hi^T P_lo^T + V_lo^T P_hi^T
// 3 x 32 cycles of v_mfma_f32_32x32x16_f16 per 16 contraction steps against 8 x 64 cycles of v_mfma_f32_32x32x2_f32 in the exact
// fp32 kernel (5.3 x fewer matrix-pipe cycles), with results at the fp32 kernel's distance from an fp64 evaluation
// (tests/test_x2_gpu.py; scripts/x2_emulation.py is the CPU model of the arithmetic).
//
// Layouts (paired, half.h): q, k [2][H][2 chunks][TP][64 physical] -- chunk c holds head dims 32c .. 32c+31 of every token as
// 32 heads | 32 tails (one 128-byte line per token and chunk), written by the QKV GEMM's fused qk-norm epilogue (gemm_h16.hip);
// v TRANSPOSED and blocked by 64 tokens, vt [H][block][2 chunks][64 d][64 physical]: chunk c holds the in-block positions
// 32c .. 32c+31 (vt_pos order) as 32 heads | 32 tails; out token-major paired (TP, 2 * H * 64) = the A operand of the out-projection.
//
// Structure = attn_h16.hip with the LDS-DMA stream (swapped products, a lane owns one query column, P fed back from the accumulator
// registers, online softmax with v_max3 row maxima and deferred rescale -- probabilities <= 2^11.5 fit fp16): every 64-key tile is
// FOUR 8 KB sub-tiles (K chunk 0 / 1, V^T chunk 0 / 1), each byte-for-byte a K or V^T tile of the 16-bit kernel (64 rows x 128 B,
// slot ^ ((row >> 1) & 7) swizzle), so a wave issues 4 LDS-DMA pieces per tile and the fragment of contraction step s lies in
// sub-tile s >> 1 at slots 2 (s & 1) + hi (heads) and 4 + 2 (s & 1) + hi (tails).  64 KB of LDS, one 8-wave block per CU.
// Few-token launches split every work item over 2 / 4 key ranges (SPLIT; attention_x2_combine_kernel merges the online-softmax partials).
#include "half.h"
#include "kernels.h"

#define XKV 64
#define XSUB (64 * 64)         // 16-bit elements of one sub-tile (8 KB)
#define XLD 72                 // row stride of the output slab (144 B)
#define X_DEFER_THR 11.5f

__device__ __forceinline__ float x_xhalf_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float x_xhalf_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float x_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// WPE = waves per SIMD the register allocation aims at: 2 (144 VGPRs, one block per CU) or 4 (128 VGPRs and 5 spilled, two blocks per CU
// -- the 64 KB of LDS allow both); tuning key 16 picks (A/B on the GPU: profiles/r05_*).
// SPLIT (few-token calls, round 5): gridDim.y blocks share the key tiles of a work item; each writes its UNNORMALISED partial O (fp32),
// its running maximum and its row sum, and attention_x2_combine_kernel merges them (online-softmax partials: O = sum_y O_y 2^((m_y - m) c) /
// sum_y l_y 2^((m_y - m) c)) and writes the head / tail planes.  One pair of 2 x 1024 points is 8 work items x 8 heads = 64 blocks for 256 CUs.
template <int WPE, bool SPLIT = false>
__global__ __launch_bounds__(512, WPE) void attention_x2_kernel(const u16* __restrict__ qk, const u16* __restrict__ vt, int vt_nblk,
                                                              u16* __restrict__ out, int TP, int heads,
                                                              const AttnWorkItem* __restrict__ items, float* __restrict__ part_o,
                                                              float* __restrict__ part_ml) {
  typedef x2_t8 T8;
  extern __shared__ __attribute__((aligned(1024))) u16 smem[];   // [2 stages][K c0 | K c1 | V c0 | V c1] = 64 KB; the 8 output slabs at the end
  const int tid = threadIdx.x;
  const int head = blockIdx.x % heads;
  const AttnWorkItem it = items[blockIdx.x / heads];
  const int len = it.seg_len;
  if (len <= 0) return;
  const int lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int seg0 = it.seg_start, seg1 = it.seg_start + len;

  const u16* Qg = qk + (size_t)head * 2 * TP * 64;                    // chunk c at + c * TP * 64
  const u16* Kg = qk + (size_t)(heads + head) * 2 * TP * 64;
  const u16* Vg = vt + (size_t)head * vt_nblk * (2 * XSUB);           // block b, chunk c at + (2 b + c) * XSUB

  const int qw0 = it.q0 + wave * 32;
  const bool wave_active = qw0 < len;   // waves beyond the segment still help stage K / V^T

  // ---- Q fragments (B operand of S^T): step s = dims 16 s + 8 hi .. +7 -> chunk s >> 1, in-chunk 16 (s & 1) + 8 hi
  T8 qh[4], ql[4];
  {
    int q = qw0 + l31;
    q = q < len ? q : len - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const u16* qp = Qg + ((size_t)(s >> 1) * TP + seg0 + q) * 64 + 16 * (s & 1) + 8 * hi;
      qh[s] = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(qp));
      ql[s] = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(qp + 32));
    }
  }

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float mrun = -1e30f, lsum = 0.f;
  const float c = 0.125f * 1.44269504088896340736f;   // 1/sqrt(64) * log2(e)

  const int b_first0 = seg0 >> 6;
  const int ntile0 = ((seg1 - 1) >> 6) - b_first0 + 1;
  // SPLIT: this block's share of the key tiles (an empty share leaves m = -1e30, l = 0, O = 0: weight 0 in the combine pass)
  const int per = SPLIT ? (ntile0 + (int)gridDim.y - 1) / (int)gridDim.y : ntile0;
  const int t_begin = SPLIT ? (int)blockIdx.y * per : 0;
  const int b_first = b_first0 + t_begin;
  const int ntile = SPLIT ? (ntile0 - t_begin < per ? ntile0 - t_begin : per) : ntile0;      // may be <= 0
  // ---- LDS-DMA: wave w stages rows 8w .. 8w+7 of each of the four sub-tiles; lane -> (row 8w + lane/8, physical slot lane%8)
  const int drow = wave * 8 + (lane >> 3);
  const int dls = ((lane & 7) ^ ((drow >> 1) & 7)) * 8;              // logical slot (in elements) this lane fetches
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) u16*)smem + (unsigned)wave * 1024u);
#define XATT_DMA1(GSRC, LDSB)                                                                                 \
  {                                                                                                           \
    unsigned keep_;                                                                                           \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(GSRC), "s"(LDSB) : "memory");                                           \
  }
#define XATT_DMA(T, BUF)                                                                                      \
  {                                                                                                           \
    const int blk_ = b_first + (T);                                                                           \
    int tok_ = blk_ * 64 + drow;                                                                              \
    tok_ = tok_ < TP ? tok_ : TP - 1;                                                                         \
    const unsigned st_ = lds_base + (unsigned)(BUF) * (4 * XSUB * 2);                                         \
    XATT_DMA1(Kg + (size_t)tok_ * 64 + dls, st_)                                                              \
    XATT_DMA1(Kg + ((size_t)TP + tok_) * 64 + dls, st_ + XSUB * 2)                                            \
    XATT_DMA1(Vg + ((size_t)(2 * blk_) * 64 + drow) * 64 + dls, st_ + 2 * XSUB * 2)                           \
    XATT_DMA1(Vg + ((size_t)(2 * blk_ + 1) * 64 + drow) * 64 + dls, st_ + 3 * XSUB * 2)                       \
  }

  if (ntile > 0) { XATT_DMA(0, 0) }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // the Q fragments have landed too -- tell the compiler (a use of every fragment), or it waits for them with vmcnt(n) inside the key
  // loop, where those waits would drain the DMA pieces of the NEXT tile it does not know about (attn_h16.hip)
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    uint4 a_ = __builtin_bit_cast(uint4, qh[s]), b_ = __builtin_bit_cast(uint4, ql[s]);
    asm volatile("" : "+v"(a_.x), "+v"(a_.y), "+v"(a_.z), "+v"(a_.w), "+v"(b_.x), "+v"(b_.y), "+v"(b_.z), "+v"(b_.w));
    qh[s] = __builtin_bit_cast(T8, a_); ql[s] = __builtin_bit_cast(T8, b_);
  }
  __syncthreads();

  const int swz = (l31 >> 1) & 7;                 // slot ^ ((row >> 1) & 7), the same for rows l31 and 32 + l31
  for (int t = 0; t < ntile; ++t) {
    const int cur = t & 1;
    const bool more = (t + 1) < ntile;
    if (more) { XATT_DMA(t + 1, cur ^ 1) }        // every wave left stage cur^1 at the last barrier

    if (wave_active) {
      const u16* stage = smem + cur * (4 * XSUB);
      // ---- S^T = K Q^T (three products): two 32-key sub-tiles x 32 queries
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const u16* kp = stage + (s >> 1) * XSUB + l31 * 64;
        const int oh = ((2 * (s & 1) + hi) ^ swz) * 8, ol = ((4 + 2 * (s & 1) + hi) ^ swz) * 8;
        const T8 k0h = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(kp + oh));
        const T8 k0l = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(kp + ol));
        const T8 k1h = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(kp + 32 * 64 + oh));
        const T8 k1l = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(kp + 32 * 64 + ol));
        s0 = H16<RAP_DT_F16>::mfma(k0l, qh[s], s0);
        s1 = H16<RAP_DT_F16>::mfma(k1l, qh[s], s1);
        s0 = H16<RAP_DT_F16>::mfma(k0h, ql[s], s0);
        s1 = H16<RAP_DT_F16>::mfma(k1h, ql[s], s1);
        s0 = H16<RAP_DT_F16>::mfma(k0h, qh[s], s0);
        s1 = H16<RAP_DT_F16>::mfma(k1h, qh[s], s1);
      }
      // ---- mask keys outside the segment (first / last tile only)
      const int tile0 = (b_first + t) * 64;
      if (tile0 < seg0 || tile0 + 64 > seg1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kg = tile0 + mfma32_crow(r, hi);
          s0[r] = (kg >= seg0 && kg < seg1) ? s0[r] : -1e30f;
          s1[r] = (kg + 32 >= seg0 && kg + 32 < seg1) ? s1[r] : -1e30f;
        }
      }
      // ---- online softmax, lane-local except one cross-half max; deferred rescale (P <= 2^X_DEFER_THR fits fp16)
      {
        float ma = x_max3(s0[0], s0[1], s0[2]), mb = x_max3(s1[0], s1[1], s1[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) { ma = x_max3(ma, s0[r], s0[r + 1]); mb = x_max3(mb, s1[r], s1[r + 1]); }
        float mx = x_max3(ma, mb, fmaxf(s0[15], s1[15]));
        mx = x_xhalf_max(mx);
        if (!__all((mx - mrun) * c <= X_DEFER_THR)) {
          const float mnew = fmaxf(mrun, mx);
          const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * c);
          mrun = mnew;
          lsum *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
      }
      {
        const f32x2 c2 = {c, c};
        const f32x2 nmc2 = {-mrun * c, -mrun * c};
        f32x2 ps2 = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          f32x2 a = {s0[2 * k], s0[2 * k + 1]};
          f32x2 b = {s1[2 * k], s1[2 * k + 1]};
          a = __builtin_elementwise_fma(a, c2, nmc2);
          b = __builtin_elementwise_fma(b, c2, nmc2);
          a.x = __builtin_amdgcn_exp2f(a.x); a.y = __builtin_amdgcn_exp2f(a.y);
          b.x = __builtin_amdgcn_exp2f(b.x); b.y = __builtin_amdgcn_exp2f(b.y);
          s0[2 * k] = a.x; s0[2 * k + 1] = a.y;
          s1[2 * k] = b.x; s1[2 * k + 1] = b.y;
          ps2 += a + b;
        }
        lsum += ps2.x + ps2.y;
      }
      // ---- O^T += V^T P^T (three products): key step ks contracts the keys held in registers 8(ks&1)..+7 of sub-tile ks>>1 =
      //      in-block positions 16 ks + 8 hi .. +7 -> V^T chunk ks >> 1, in-chunk 16 (ks & 1) + 8 hi
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int rb = 8 * (ks & 1);
        f32x8 p8;
        if ((ks >> 1) == 0) p8 = f32x8{s0[rb + 0], s0[rb + 1], s0[rb + 2], s0[rb + 3], s0[rb + 4], s0[rb + 5], s0[rb + 6], s0[rb + 7]};
        else p8 = f32x8{s1[rb + 0], s1[rb + 1], s1[rb + 2], s1[rb + 3], s1[rb + 4], s1[rb + 5], s1[rb + 6], s1[rb + 7]};
        T8 ph, pl;
        x2_split8_nosat(p8, ph, pl);
        const u16* vp = stage + (2 + (ks >> 1)) * XSUB + l31 * 64;
        const int oh = ((2 * (ks & 1) + hi) ^ swz) * 8, ol = ((4 + 2 * (ks & 1) + hi) ^ swz) * 8;
        const T8 v0h = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(vp + oh));
        const T8 v0l = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(vp + ol));
        const T8 v1h = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(vp + 32 * 64 + oh));
        const T8 v1l = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(vp + 32 * 64 + ol));
        o0 = H16<RAP_DT_F16>::mfma(v0l, ph, o0);
        o1 = H16<RAP_DT_F16>::mfma(v1l, ph, o1);
        o0 = H16<RAP_DT_F16>::mfma(v0h, pl, o0);
        o1 = H16<RAP_DT_F16>::mfma(v1h, pl, o1);
        o0 = H16<RAP_DT_F16>::mfma(v0h, ph, o0);
        o1 = H16<RAP_DT_F16>::mfma(v1h, ph, o1);
      }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's four pieces of tile t + 1 have landed
    __syncthreads();
  }

  if (!wave_active) return;
  if constexpr (SPLIT) {
    // partial results of this key range: O^T unnormalised (relative to mrun), row-major (TP, heads * 64) fp32 per range; (m, l) per (token, head)
    const int q = qw0 + l31;
    if (q < len) {
      const size_t tok = (size_t)blockIdx.y * TP + (size_t)(seg0 + q);
      float* po = part_o + tok * ((size_t)heads * 64) + head * 64 + 4 * hi;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(po + 8 * g) = float4{o0[4 * g + 0], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]};
        *reinterpret_cast<float4*>(po + 32 + 8 * g) = float4{o1[4 * g + 0], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]};
      }
    }
    const float lall = x_xhalf_sum(lsum);
    if (q < len && hi == 0) *reinterpret_cast<float2*>(part_ml + (((size_t)blockIdx.y * TP + (size_t)(seg0 + q)) * heads + head) * 2) = float2{mrun, lall};
    return;
  }
  // ---- normalise, split and store.  Lane owns query l31; register r of tile e is head dim 32 e + crow(r, hi) (groups of 4 contiguous
  // dims), i.e. tile e IS chunk e of this head.  Every wave is past the last tile's barrier: the stages are free.  Slab of this wave:
  // [32 queries][72]: 32 heads | 32 tails of one chunk, written and drained once per chunk (LDS operations of a wave execute in order).
  const float inv = 1.0f / x_xhalf_sum(lsum);
  u16* slab = smem + wave * (32 * XLD);
  u16* wp = slab + l31 * XLD + 4 * hi;
  const size_t orow = (size_t)heads * 128;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v4 = e == 0 ? f32x4{o0[4 * g + 0], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]} : f32x4{o1[4 * g + 0], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]};
      const f32x4 n4 = v4 * inv;
      const x2_t4 h4 = __builtin_convertvector(n4, x2_t4);
      const x2_t4 l4 = __builtin_convertvector(n4 - __builtin_convertvector(h4, f32x4), x2_t4);
      *reinterpret_cast<uint2*>(wp + 8 * g) = __builtin_bit_cast(uint2, h4);
      *reinterpret_cast<uint2*>(wp + 32 + 8 * g) = __builtin_bit_cast(uint2, l4);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (lane >> 3) + 8 * i, piece = lane & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(slab + row * XLD + piece * 8);
      if (qw0 + row < len)
        *reinterpret_cast<uint4*>(out + (size_t)(seg0 + qw0 + row) * orow + head * 128 + e * 64 + piece * 8) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slab's reads have returned before chunk 1 overwrites it
    __builtin_amdgcn_wave_barrier();
  }
}

// Measured and NOT kept (round 5, GPU call 2; source at commit "split-precision attention: software-pipelined kernel", numbers in
// profiles/r05_c2_x2_attention_pipelined_ab.jsonl): a software-pipelined form of this kernel -- the softmax of tile t + 1 interleaved, by
// sched_group_barrier, with the 24 P V MFMAs of tile t (K one tile ahead of V^T in the LDS ring, ping-pong P fragments, 200 VGPRs) so
// that a wave never stops issuing MFMAs for the ~180 VALU instructions of a tile's softmax: 5.99 / 11.44 ms per launch at L = 4096 /
// 8192 against 5.62 / 11.05 ms for this kernel, 43.4 k vs 45.1 k points/s for the whole call.  Two blocks per CU (WPE = 4) are no
// faster either.  As for the bf16 kernel, every schedule lands on the same ~1.2 PFLOP/s of issued fp16 MFMA work: the
// SQ counters (profiles/r05_c3_mfma_utilisation_x2.txt) show the clock the part sustains under this load, not the schedule, as the limit.
// Round 6 (GPU call 19, profiles/r06_c19_x2_attention_ring_latency.jsonl): the four-stage K / V^T ring that took the 16-bit kernel's few-token
// launches from 24.1 to 19.2 us (attn_h16.hip) was built for the SPLIT launches of this kernel too (128 KB of LDS, counted vmcnt, bit-identical)
// and changed nothing: 25.5 vs 25.4 ms per configs[0]-geometry call, 80.8 vs 80.5 at 4 096 rows.  A tile here is 48 MFMAs, ~350 VALU
// instructions and 32 ds_read_b128 per wave with all eight waves of the block working -- about 1 us each of matrix pipe, VALU and LDS
// bandwidth per tile and CU, executed phase by phase by waves that one barrier per tile keeps in step -- not the round trip of the next tile.
rap_tuning_t g_rap_attn_x2_wpe = 2;      // tuning key 16: 2 / 4 = one / two blocks per CU
extern rap_tuning_t g_rap_attn_split;    // attn_f32.hip, tuning key 5: split few-token attention launches over key ranges (1, default) or not (0)

// one thread per (token, 8 consecutive head dims): merge the key-range partials, normalise, write the head / tail planes
__global__ __launch_bounds__(256) void attention_x2_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                                   u16* __restrict__ out, int TP, int n_tokens, int heads, int splits,
                                                                   const int32_t* __restrict__ cover_cu, int cover_nseg) {
  const int dmodel = heads * 64;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)n_tokens * (dmodel / 8)) return;
  const long t = i / (dmodel / 8);
  const int c = (int)(i % (dmodel / 8)) * 8;                  // logical column: head = c >> 6, dims (c & 63) .. +7 (inside one 32-dim chunk)
  const int head = c >> 6;
  u16* dst = out + (size_t)t * (heads * 128) + head * 128 + ((c & 63) >> 5) * 64 + (c & 31);
  // a token outside every segment (an inconsistent table whose sanitised copy ends below n_tokens) has no partials: zeros, as the
  // unsplit kernel leaves such rows (the stale bytes of the aliased buffers used to be normalised by 1 / l = 1 / garbage)
  if (cover_cu && (t < cover_cu[0] || t >= cover_cu[cover_nseg])) {
    *reinterpret_cast<uint4*>(dst) = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(dst + 32) = make_uint4(0, 0, 0, 0);
    return;
  }
  const float cexp = 0.125f * 1.44269504088896340736f;
  float m = -1e30f;
  for (int s = 0; s < splits; ++s) m = fmaxf(m, part_ml[(((size_t)s * TP + t) * heads + head) * 2]);
  float l = 0.f;
  f32x8 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < splits; ++s) {
    const float2 ml = *reinterpret_cast<const float2*>(part_ml + (((size_t)s * TP + t) * heads + head) * 2);
    const float w = __builtin_amdgcn_exp2f((ml.x - m) * cexp);       // an empty range: m_y = -1e30 -> weight 0 (and l_y = 0, O_y = 0)
    l += ml.y * w;
    const float4 a0 = *reinterpret_cast<const float4*>(part_o + ((size_t)s * TP + t) * dmodel + c);
    const float4 a1 = *reinterpret_cast<const float4*>(part_o + ((size_t)s * TP + t) * dmodel + c + 4);
    acc += f32x8{a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w} * w;
  }
  const float inv = l > 0.f ? 1.0f / l : 0.f;      // (every key range of a covered token saw at least one key: l > 0 there)
  x2_t8 h8, l8;
  x2_split8_nosat(acc * inv, h8, l8);
  *reinterpret_cast<uint4*>(dst) = __builtin_bit_cast(uint4, h8);
  *reinterpret_cast<uint4*>(dst + 32) = __builtin_bit_cast(uint4, l8);
}

// key ranges per work item of a few-token launch (1 = no split): work lists that leave most of the 256 one-block-per-CU slots empty
int attention_x2_splits(int max_items, int heads) {
  if (g_rap_attn_split == 0) return 1;
  const long blocks = (long)max_items * heads;
  return blocks <= 96 ? 4 : blocks <= 192 ? 2 : 1;
}

// part_o: splits x TP x heads*64 floats, part_ml: splits x TP x heads x 2 floats (splits > 1 only)
int launch_attention_x2(hipStream_t stream, const u16* qk, const u16* vt, int vt_nblk, u16* out, int TP, int heads,
                        const AttnWorkItem* items, int max_items, float* part_o, float* part_ml, int splits, int n_tokens,
                        const int32_t* cover_cu, int cover_nseg) {
  if (max_items <= 0 || TP <= 0) return RAP_OK;
  if (heads <= 0 || vt_nblk * 64 < TP) return RAP_ERR_INVALID;
  constexpr int LDS = 2 * 4 * XSUB * 2;      // 64 KB
  static_assert(8 * 32 * XLD * 2 <= LDS, "output slabs must fit the stages");
  if (splits > 1) {
    if (!part_o || !part_ml) return RAP_ERR_INVALID;
    const void* fn = reinterpret_cast<const void*>(attention_x2_kernel<2, true>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      rap_set_last_hip_error((int)hipGetLastError());
      return RAP_ERR_HIP;
    }
    hipLaunchKernelGGL((attention_x2_kernel<2, true>), dim3(max_items * heads, splits), dim3(512), LDS, stream, qk, vt, vt_nblk, out, TP, heads, items,
                       part_o, part_ml);
    RAP_LAUNCH_CHECK();
    // only the rows that belong to a segment were written by a block: the filler rows of the padded buffers (n_tokens .. TP - 1) keep
    // the zeros prepare_static put there (they must stay finite: masked keys multiply their V column by p = 0)
    const int nt = n_tokens > 0 && n_tokens < TP ? n_tokens : TP;
    const long n8 = (long)nt * heads * 8;
    hipLaunchKernelGGL(attention_x2_combine_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, stream, part_o, part_ml, out, TP, nt, heads, splits,
                       cover_cu, cover_nseg);
    RAP_LAUNCH_CHECK();
    return RAP_OK;
  }
  float* const no = nullptr;
  const bool two = g_rap_attn_x2_wpe == 4;
  const void* fn = two ? reinterpret_cast<const void*>(attention_x2_kernel<4>) : reinterpret_cast<const void*>(attention_x2_kernel<2>);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
    rap_set_last_hip_error((int)hipGetLastError());
    return RAP_ERR_HIP;
  }
  if (two) hipLaunchKernelGGL(attention_x2_kernel<4>, dim3(max_items * heads), dim3(512), LDS, stream, qk, vt, vt_nblk, out, TP, heads, items, no, no);
  else hipLaunchKernelGGL(attention_x2_kernel<2>, dim3(max_items * heads), dim3(512), LDS, stream, qk, vt, vt_nblk, out, TP, heads, items, no, no);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
