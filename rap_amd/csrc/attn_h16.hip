// Variable-length, non-causal softmax attention on the bf16 / fp16 matrix cores (fp32 softmax state and accumulation).
//
// Reduced-precision twin of attn_f32.hip; replaces flash_attn.flash_attn_varlen_qkvpacked_func as the reference
// runs it on a GPU (fp16 / bf16 `attn_dtype`, flow_model/layer.py:106-111 per part, :123-128 per sample).
//
// Layout: q,k head-major [2][H][TP][64] 16-bit (written by the QKV GEMM epilogue, normalised in place by qknorm);
// v TRANSPOSED and blocked by 64 tokens, vt[H][block][64 d][64 pos] with pos = vt_pos(token & 63) (half.h);
// out token-major (TP, H*64) 16-bit = the A operand of the out-projection GEMM.
//
// Design (gfx950 only; Dh = 64):
//  * block = 256 queries of one (segment, head): 8 waves x 32 queries; K / V^T streamed in 64-key tiles through
//    double-buffered LDS (global -> registers -> LDS; loads of tile t+1 are issued before the MFMAs of tile t and
//    parked after them: one barrier per tile).  Key tiles are the GLOBAL 64-token blocks that intersect the
//    segment (the V^T image is blocked that way); keys outside the segment are masked, so ragged segments cost
//    at most one extra tile.
//  * "swapped" products on v_mfma_f32_32x32x16_{bf16,f16}:
//      S^T (key x query) = K (key x d) * Q^T (d x query)     A = K rows from LDS (one ds_read_b128 = 8 d), B = Q in VGPRs
//      O^T (d x query)   = V^T (d x key) * P^T (key x query)  A = V^T rows from LDS (one ds_read_b128 = 8 keys), B = P
//    A lane owns ONE query column and 16 of every 32 keys, so the online-softmax state is lane-local (one
//    lane^32 exchange per tile for the row maximum).
//  * No data movement between the two products: the 8 accumulator registers 8(s&1)..8(s&1)+7 of S^T sub-tile s>>1
//    hold the keys 16s + 4hi + {0..3} and 16s + 8 + 4hi + {0..3}; vt_pos stores exactly those keys contiguously,
//    so P is converted to 16 bit in place (v_cvt_pk) and fed back as the B operand: no LDS round trip, no shuffles.
//  * LDS rows are 144 bytes (9 sixteen-byte slots): the 16 rows of each ds_read_b128 lane group hit 16 distinct slots.
//  * softmax scale * log2(e) is applied inside the exponent's FMA (fp32 scores), exponentials are v_exp_f32.
//  * grid = work items x heads with head = blockIdx % H (= the XCD: each XCD's L2 serves one head's K/V).
#include "half.h"
#include "kernels.h"

#define HKV 64
#define HLD 72   // LDS row stride in 16-bit elements (144 B)

__device__ __forceinline__ float h_xhalf_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float h_xhalf_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ABL: timing-only ablations (scripts/kernel_bench.py --h16-attn-variant; results are NOT attention): bit 0 = no softmax
// VALU (P = raw S), bit 1 = no K/V streaming (every tile re-uses the first one: no global loads, LDS writes, barriers),
// bit 3 = no transcendental, bit 4 = no O rescale, bit 5 = no row-maximum chain.
// OPT: bit 0 = row maximum through v_max3_f32, bit 1 = deferred rescale (threshold DEFER_THR in the base-2 exponent),
// bit 2 = s_setprio(1) around the MFMA clusters (measured: no gain), bit 3 = bounded softmax (see below).
#define DEFER_THR 11.5f   // = 8 in natural-log units: P <= e^8
__device__ __forceinline__ float hmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// PERSIST (r02, rap_set_tuning(3, 19)): the grid is a fixed number of blocks that walk the (work item, head) list with stride gridDim.x
// (a multiple of the head count, so a block keeps its head = its XCD) instead of one block per entry.
// ROT (r02, rap_set_tuning(3, 20)): block j of a segment starts its walk over the key tiles at ITS OWN diagonal (tile 4j) and wraps, so
// that the blocks of one (segment, head) -- which run at the same time on one XCD -- do not all ask for the same cold K / V^T tile
// at the same moment (softmax is order-independent; the masks follow the rotated tile index).
#ifdef RAP_ABLATION_BUILD
__device__ unsigned long long* g_attn_ts = nullptr;      // [block][8] s_memtime stamps of wave 0 (ABL bit 8), set by rap_debug_attn_ts
extern "C" int rap_debug_attn_ts(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_ts), &p, sizeof(p)) == hipSuccess ? 0 : -3; }
#define ATT_TS(I) if ((ABL & 256) && g_attn_ts && threadIdx.x == 0) { g_attn_ts[(size_t)blockIdx.x * 8 + (I)] = __builtin_readcyclecounter(); \
    if ((I) == 0) g_attn_ts[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime(); if ((I) == 5) g_attn_ts[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime(); }
#else
#define ATT_TS(I)
#endif
// LST (r02, rap_set_tuning(3, 22)): the output tile leaves through the wave's own 4.6 KB slab of the (by then idle) K / V^T buffers and
// is stored as whole 128-byte rows, 16 bytes per lane, instead of eight 8-byte pieces per lane at a 1 KB row stride.
// W16 (r02, built at the end of the round, UNMEASURED; rap_set_tuning(3, 24)): 16 waves x 32 queries = 512 queries per block on a work
// list of 512-query items: the K / V^T stream, its staging instructions and the barrier are shared by twice the matrix work (threads
// 0-511 stage K, 512-1023 stage V^T: one 16-byte chunk per thread and tile).  Same 4 waves per SIMD, one block per CU.
template <int DT, int ABL, int OPT, bool PERSIST = false, bool ROT = false, bool LST = false, bool W16 = false>
__global__ __launch_bounds__(W16 ? 1024 : 512, 2) void attention_h16_kernel(const u16* __restrict__ qk, const u16* __restrict__ vt,
                                                               int vt_nblk, u16* __restrict__ out, int TP, int heads,
                                                               const AttnWorkItem* __restrict__ items,
                                                               const float* __restrict__ bound, int total_blocks) {
  typedef typename H16<DT>::T8 T8;
  __shared__ __attribute__((aligned(16))) u16 smem[4 * HKV * HLD];
  u16* Ks = smem;                    // [2][64 keys][72]
  u16* Vs = smem + 2 * HKV * HLD;    // [2][64 d][72]   (columns = vt_pos of the key)

  const int tid = threadIdx.x;
  int vb = blockIdx.x;
  ATT_TS(0)
  do {
  const int head = vb % heads;
  const AttnWorkItem it = items[vb / heads];
  const int len = it.seg_len;
  if (len <= 0) continue;
  const int lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int seg0 = it.seg_start, seg1 = it.seg_start + len;

  const u16* Qg = qk + (size_t)head * TP * 64;
  const u16* Kg = qk + (size_t)(heads + head) * TP * 64;
  const u16* Vg = vt + (size_t)head * vt_nblk * (64 * 64);

  const int qw0 = it.q0 + wave * 32;
  const bool wave_active = qw0 < len;   // waves beyond the segment still help stage K/V

  // ---- Q fragments (B operand of S^T): qf[s] = Q[q][16s + 8hi .. +7]
  T8 qf[4];
  {
    int q = qw0 + l31;
    q = q < len ? q : len - 1;
    const u16* qp = Qg + (size_t)(seg0 + q) * 64 + 8 * hi;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (ABL & 128) qf[s] = __builtin_bit_cast(T8, make_uint4(0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u + s, 0x3c003c00u));   // timing-only: no Q loads
      else qf[s] = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(qp + 16 * s));
    }
  }

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  // OPT bit 3: bounded softmax -- the caller guarantees q.k/8 <= bound[head] (after qk-norm: 8 max|gamma_q| max|gamma_k|), so
  // the fixed offset bound replaces the running maximum: p = exp(s - bound), no max chain, no rescale, no branch.
  // OPT bit 4 (with bit 3, bf16 only): q arrives pre-scaled by log2(e)/8 (qknorm_h16, q_mul = log2 e), so a score IS the exp2
  // argument, and because |score| <= bound * log2(e) <= 58 the offset is dropped altogether: p = exp2(score) lies in
  // [2^-58, 2^58], far inside the range of fp32 and bf16, and softmax is invariant to the common factor.  No per-score FMA:
  // 848 -> 917 TF per part, 952 -> 1032 per sample (r01 run 43).  Measured and rejected on top of it (run 44): row sums from the
  // matrix pipe (a third O tile against an all-ones V^T: 4 more MFMAs instead of 16 v_pk_add_f32 per key tile) -> 903 / 969.
  float mrun = (OPT & 16) ? 0.f : (OPT & 8) ? bound[head] * 8.0f : -1e30f, lsum = 0.f;
  const float c = 0.125f * 1.44269504088896340736f;   // 1/sqrt(64) * log2(e)

  // ---- staging: 512 threads, one 16-byte chunk of K and one of V^T per thread per tile (W16: 1024 threads, K or V^T)
  const int stid = W16 ? (tid & 511) : tid;
  const bool st_k = !W16 || tid < 512, st_v = !W16 || tid >= 512;
  const int srow = stid >> 3, sch = (stid & 7) * 8;
  const int b_first = seg0 >> 6;
  const int ntile = ((seg1 - 1) >> 6) - b_first + 1;
  const int soff = srow * HLD + sch;
  uint4 rk = make_uint4(0, 0, 0, 0), rv = rk;
#define HATT_LOAD(T)                                                                                 \
  {                                                                                                  \
    const int blk_ = b_first + (T);                                                                  \
    int tok_ = blk_ * 64 + srow;                                                                     \
    tok_ = tok_ < TP ? tok_ : TP - 1;                                                                \
    if (st_k) rk = *reinterpret_cast<const uint4*>(Kg + (size_t)tok_ * 64 + sch);                    \
    if (st_v) rv = *reinterpret_cast<const uint4*>(Vg + ((size_t)blk_ * 64 + srow) * 64 + sch);      \
  }
#define HATT_STORE(BUF)                                                                              \
  if (st_k) *reinterpret_cast<uint4*>(Ks + (BUF) * (HKV * HLD) + soff) = rk;                         \
  if (st_v) *reinterpret_cast<uint4*>(Vs + (BUF) * (HKV * HLD) + soff) = rv;

  int rt = ROT ? (int)(((unsigned)it.q0 >> 6) % (unsigned)ntile) : 0;     // rotated tile index of iteration t
  ATT_TS(1)
  HATT_LOAD(rt)
  HATT_STORE(0)
  __syncthreads();
  ATT_TS(2)
  if (ABL & 256) { const uint4 q0_ = __builtin_bit_cast(uint4, qf[0]); const uint4 q3_ = __builtin_bit_cast(uint4, qf[3]); if ((q0_.x ^ q3_.w) == 0x9e3779b9u && tid == 9999) return; }   // the Q loads complete before stamp 3
  ATT_TS(3)

  for (int t = 0; t < ntile; ++t) {
    const int cur = (ABL & 2) ? 0 : (t & 1);
    const bool more = (ABL & 2) ? false : (t + 1) < ntile;
    const int rt_cur = rt;
    if (ROT) { rt = rt + 1; rt = rt == ntile ? 0 : rt; }
    if (more) { HATT_LOAD(ROT ? rt : t + 1) }

    if (wave_active) {
      // ---- S^T = K Q^T : two 32-key sub-tiles x 32 queries
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
      const u16* kp = Ks + cur * (HKV * HLD) + l31 * HLD + 8 * hi;
      if (OPT & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const T8 k0 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(kp + 16 * s));
        const T8 k1 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(kp + 32 * HLD + 16 * s));
        s0 = H16<DT>::mfma(k0, qf[s], s0);
        s1 = H16<DT>::mfma(k1, qf[s], s1);
      }
      if (OPT & 4) __builtin_amdgcn_s_setprio(0);
      // ---- mask keys outside the segment (first / last tile only)
      const int tile0 = (b_first + (ROT ? rt_cur : t)) * 64;
      if (tile0 < seg0 || tile0 + 64 > seg1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kg = tile0 + mfma32_crow(r, hi);
          s0[r] = (kg >= seg0 && kg < seg1) ? s0[r] : -1e30f;
          s1[r] = (kg + 32 >= seg0 && kg + 32 < seg1) ? s1[r] : -1e30f;
        }
      }
      // ---- online softmax, lane-local except one cross-half max
      if (!(ABL & 1)) {
        float mx;
        if (OPT & 8) {
          mx = mrun;                                  // bounded: nothing to track
        } else if (ABL & 32) {
          mx = 8.0f;                                  // timing-only: no row-maximum chain
        } else if (OPT & 1) {
          float ma = hmax3(s0[0], s0[1], s0[2]), mb = hmax3(s1[0], s1[1], s1[2]);   // two independent v_max3_f32 chains
#pragma unroll
          for (int r = 3; r < 15; r += 2) { ma = hmax3(ma, s0[r], s0[r + 1]); mb = hmax3(mb, s1[r], s1[r + 1]); }
          mx = hmax3(ma, mb, fmaxf(s0[15], s1[15]));
          mx = h_xhalf_max(mx);
        } else {
          mx = fmaxf(s0[0], s1[0]);
#pragma unroll
          for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
          mx = h_xhalf_max(mx);
        }
        if (OPT & 8) {
        } else if (OPT & 2) {
          // deferred rescale: keep the running maximum as long as no row of the wave grew by more than DEFER_THR in the
          // exponent (P <= 2^DEFER_THR: no overflow in bf16 / fp16 / the fp32 sums); O and l are rescaled only then.
          // Textbook order: the decision precedes this tile's exponentials and follows the previous tile's P*V.
          if (!__all((mx - mrun) * c <= DEFER_THR)) {
            const float mnew = fmaxf(mrun, mx);
            const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * c);
            mrun = mnew;
            lsum *= alpha;
            if (!(ABL & 16)) {
#pragma unroll
              for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            }
          }
        } else {
          const float mnew = fmaxf(mrun, mx);
          const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * c);
          mrun = mnew;
          lsum *= alpha;
          if (!(ABL & 16)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
          }
        }
        // exponent FMA and the row sum on the packed fp32 pipe (v_pk_fma_f32 / v_pk_add_f32: two lanes' worth per issue
        // slot): the kernel is VALU-issue bound (PMC: ~10 VALU per MFMA at ~4.8 cycles each vs 32 cycles per MFMA).
        const f32x2 c2 = {c, c};
        const f32x2 nmc2 = {-mrun * c, -mrun * c};
        f32x2 ps2 = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          f32x2 a = {s0[2 * k], s0[2 * k + 1]};
          f32x2 b = {s1[2 * k], s1[2 * k + 1]};
          if (!(OPT & 16)) {
            a = __builtin_elementwise_fma(a, c2, nmc2);
            b = __builtin_elementwise_fma(b, c2, nmc2);
          }
          if (ABL & 8) {                              // timing-only: no transcendental
            a *= 0.001f; b *= 0.001f;
          } else {
            a.x = __builtin_amdgcn_exp2f(a.x); a.y = __builtin_amdgcn_exp2f(a.y);
            b.x = __builtin_amdgcn_exp2f(b.x); b.y = __builtin_amdgcn_exp2f(b.y);
          }
          s0[2 * k] = a.x; s0[2 * k + 1] = a.y;
          s1[2 * k] = b.x; s1[2 * k + 1] = b.y;
          ps2 += a + b;
        }
        lsum += ps2.x + ps2.y;
      } else { lsum += s0[0]; }
      // ---- O^T += V^T P^T : key step ks contracts the keys held in registers 8(ks&1)..+7 of sub-tile ks>>1
      const u16* vp = Vs + cur * (HKV * HLD) + l31 * HLD + 8 * hi;
      if (OPT & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int rb = 8 * (ks & 1);
        T8 pb;
        if ((ks >> 1) == 0)
          pb = h16_pack8<DT>(s0[rb + 0], s0[rb + 1], s0[rb + 2], s0[rb + 3], s0[rb + 4], s0[rb + 5], s0[rb + 6], s0[rb + 7]);
        else
          pb = h16_pack8<DT>(s1[rb + 0], s1[rb + 1], s1[rb + 2], s1[rb + 3], s1[rb + 4], s1[rb + 5], s1[rb + 6], s1[rb + 7]);
        const T8 v0 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(vp + 16 * ks));
        const T8 v1 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(vp + 32 * HLD + 16 * ks));
        o0 = H16<DT>::mfma(v0, pb, o0);
        o1 = H16<DT>::mfma(v1, pb, o1);
      }
      if (OPT & 4) __builtin_amdgcn_s_setprio(0);
    }

    if (more) { HATT_STORE(cur ^ 1) }
    if (!(ABL & 2)) __syncthreads();
  }

  ATT_TS(4)
  if (!wave_active) continue;
  // ---- normalise and store: lane owns query l31; register r of tile e is d = 32e + crow(r, hi): groups of 4 contiguous d
  const int q = qw0 + l31;
  const float inv = 1.0f / h_xhalf_sum(lsum);
  if ((ABL & 64) && inv != 12345.678f) continue;    // timing-only: no output stores
  static_assert(!(LST && W16), "the row-store slabs are sized for 8 waves");
  if (LST) {
    // every wave of the block is past the last tile's barrier: the K / V^T buffers are free.  Slab of this wave: [32 queries][72]
    u16* slab = smem + wave * (32 * HLD);
    u16* wp = slab + l31 * HLD + 4 * hi;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *reinterpret_cast<uint2*>(wp + 8 * g) =
          h16_pack4<DT>(o0[4 * g + 0] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
      *reinterpret_cast<uint2*>(wp + 32 + 8 * g) =
          h16_pack4<DT>(o1[4 * g + 0] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slab is private to the wave and LDS operations of a wave execute in order:
    __builtin_amdgcn_wave_barrier();                          // no block barrier (waves without queries have left already)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (lane >> 3) + 8 * i, piece = lane & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(slab + row * HLD + piece * 8);
      if (qw0 + row < len)
        *reinterpret_cast<uint4*>(out + (size_t)(seg0 + qw0 + row) * (heads * 64) + head * 64 + piece * 8) = v;
    }
    if (PERSIST) __syncthreads();
  } else if (q < len) {
    u16* op = out + (size_t)(seg0 + q) * (heads * 64) + head * 64 + 4 * hi;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *reinterpret_cast<uint2*>(op + 8 * g) =
          h16_pack4<DT>(o0[4 * g + 0] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
      *reinterpret_cast<uint2*>(op + 32 + 8 * g) =
          h16_pack4<DT>(o1[4 * g + 0] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
    }
  }
  ATT_TS(5)
  } while (PERSIST && (vb += (int)gridDim.x) < total_blocks);
}

// ---------------------------------------------------------------------------------------------
// "ping-pong" schedule (rap_set_tuning(3, 11); NOT the default): the two waves that share a SIMD (w and w+4 of the 8-wave block) run the SAME tile loop half an
// iteration apart, held in anti-phase by two barriers per key tile:
//
//     phase p     :  waves 0-3  M(i): S(i) = K(i) Q^T, O += V(i-1) P(i-1)   16 MFMA   |  waves 4-7  V(i-1): softmax -> P(i-1)
//     phase p + 1 :  waves 0-3  V(i): softmax of S(i) -> P(i)               ~100 VALU |  waves 4-7  M(i)
//
// so on every SIMD one wave is in its matrix segment while its partner is in its VALU segment (MI355X_MICROARCH.md "Two
// waves per SIMD"): the matrix pipe and the vector pipe are both busy all the time instead of taking turns, which is
// what v1 was suspected of (all waves of a block leave the barrier in the same segment).  MEASURED on MI355X: 800 TF vs
// v1's 940 -- the SIMD's issue port, not phase alignment, is the limit (PMC: ~10 VALU per MFMA at ~4.8 issue cycles each
// + 8 per MFMA ~ 55 issue cycles per 32-cycle MFMA), and the second barrier per tile costs more than the anti-phase
// buys.  Kept selectable because it is the cleanest A/B for that question.  Works with the online softmax
// (the rescale branch lives in the V segment) and with the bounded one; one query tile per wave, ~128 VGPRs, two blocks
// per CU.  K(j+1) and V(j) are fetched in phase 2j (global -> registers), parked in LDS in phase 2j+1 and first read in
// phase 2j+2; K and V are double-buffered.
// ---------------------------------------------------------------------------------------------
template <int DT, bool BOUNDED>
__global__ __launch_bounds__(512, 4) void attention_h16_pp_kernel(const u16* __restrict__ qk, const u16* __restrict__ vt,
                                                                  int vt_nblk, u16* __restrict__ out, int TP, int heads,
                                                                  const AttnWorkItem* __restrict__ items,
                                                                  const float* __restrict__ bound) {
  typedef typename H16<DT>::T8 T8;
  __shared__ __attribute__((aligned(16))) u16 smem[4 * HKV * HLD];
  u16* Ks = smem;                    // [2][64 keys][72]
  u16* Vs = smem + 2 * HKV * HLD;    // [2][64 d][72]

  const int head = blockIdx.x % heads;
  const AttnWorkItem it = items[blockIdx.x / heads];
  const int len = it.seg_len;
  if (len <= 0) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int grp = wave >> 2;         // 0: leads, 1: trails by one phase
  const int seg0 = it.seg_start, seg1 = it.seg_start + len;

  const u16* Qg = qk + (size_t)head * TP * 64;
  const u16* Kg = qk + (size_t)(heads + head) * TP * 64;
  const u16* Vg = vt + (size_t)head * vt_nblk * (64 * 64);

  // waves w and w+4 share a SIMD; give them ADJACENT query tiles so that a block still covers 256 consecutive queries
  const int qw0 = it.q0 + ((wave & 3) * 2 + grp) * 32;
  const bool wave_active = qw0 < len;

  T8 qf[4];
  {
    int q = qw0 + l31;
    q = q < len ? q : len - 1;
    const u16* qp = Qg + (size_t)(seg0 + q) * 64 + 8 * hi;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(qp + 16 * s));
  }
  f32x16 o0, o1, s0, s1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; s0[r] = 0.f; s1[r] = 0.f; }
  T8 pb[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) pb[ks] = (T8)0;
  const float c = 0.125f * 1.44269504088896340736f;
  float mrun = BOUNDED ? bound[head] * 8.0f : -1e30f;     // BOUNDED: the fixed exponent offset (scores are q.k, bound is on q.k/8)
  f32x2 l2 = {0.f, 0.f};

  const int srow = tid >> 3, sch = (tid & 7) * 8;
  const int b_first = seg0 >> 6;
  const int ntile = ((seg1 - 1) >> 6) - b_first + 1;
  const int soff = srow * HLD + sch;
  uint4 rk, rv;
  {  // K(0)
    int tok = b_first * 64 + srow;
    tok = tok < TP ? tok : TP - 1;
    rk = *reinterpret_cast<const uint4*>(Kg + (size_t)tok * 64 + sch);
    *reinterpret_cast<uint4*>(Ks + soff) = rk;
  }
  __syncthreads();

  const int nphase = 2 * ntile + 2;
  for (int p = 0; p < nphase; ++p) {
    const int j = p >> 1;                       // staging pair j = {K(j+1), V(j)}
    if ((p & 1) == 0 && j < ntile) {
      const int blk = b_first + j;
      int tok = (blk + 1) * 64 + srow;
      tok = tok < TP ? tok : TP - 1;
      rk = *reinterpret_cast<const uint4*>(Kg + (size_t)tok * 64 + sch);          // K(j+1) (a harmless over-read after the last tile)
      rv = *reinterpret_cast<const uint4*>(Vg + ((size_t)blk * 64 + srow) * 64 + sch);
    }
    const int lp = p - grp;
    if (wave_active && lp >= 0 && lp <= 2 * ntile) {
      const int i = lp >> 1;
      if ((lp & 1) == 0) {
        // ---------------- M segment: S(i) = K(i) Q^T and O += V(i-1) P(i-1), four independent accumulator chains
        const u16* kp = Ks + (i & 1) * (HKV * HLD) + l31 * HLD + 8 * hi;
        const u16* vp = Vs + ((i + 1) & 1) * (HKV * HLD) + l31 * HLD + 8 * hi;
        if (i < ntile && i >= 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const T8 k0 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(kp + 16 * s));
            const T8 k1 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(kp + 32 * HLD + 16 * s));
            const T8 v0 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(vp + 16 * s));
            const T8 v1 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(vp + 32 * HLD + 16 * s));
            s0 = H16<DT>::mfma(k0, qf[s], s0);
            o0 = H16<DT>::mfma(v0, pb[s], o0);
            s1 = H16<DT>::mfma(k1, qf[s], s1);
            o1 = H16<DT>::mfma(v1, pb[s], o1);
          }
        } else if (i < ntile) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const T8 k0 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(kp + 16 * s));
            const T8 k1 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(kp + 32 * HLD + 16 * s));
            s0 = H16<DT>::mfma(k0, qf[s], s0);
            s1 = H16<DT>::mfma(k1, qf[s], s1);
          }
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const T8 v0 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(vp + 16 * s));
            const T8 v1 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(vp + 32 * HLD + 16 * s));
            o0 = H16<DT>::mfma(v0, pb[s], o0);
            o1 = H16<DT>::mfma(v1, pb[s], o1);
          }
        }
      } else {
        // ---------------- V segment: softmax of S(i) -> P(i)
        const int tile0 = (b_first + i) * 64;
        if (tile0 < seg0 || tile0 + 64 > seg1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kg = tile0 + mfma32_crow(r, hi);
            s0[r] = (kg >= seg0 && kg < seg1) ? s0[r] : -1e30f;
            s1[r] = (kg + 32 >= seg0 && kg + 32 < seg1) ? s1[r] : -1e30f;
          }
        }
        if (!BOUNDED) {
          float ma = hmax3(s0[0], s0[1], s0[2]), mb = hmax3(s1[0], s1[1], s1[2]);
#pragma unroll
          for (int r = 3; r < 15; r += 2) { ma = hmax3(ma, s0[r], s0[r + 1]); mb = hmax3(mb, s1[r], s1[r + 1]); }
          float mx = hmax3(ma, mb, fmaxf(s0[15], s1[15]));
          mx = h_xhalf_max(mx);
          if (!__all((mx - mrun) * c <= DEFER_THR)) {          // the previous tile's P*V (M segment) is complete here
            const float mnew = fmaxf(mrun, mx);
            const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * c);
            mrun = mnew;
            l2 *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
          }
        }
        const f32x2 c2 = {c, c};
        const f32x2 nmc2 = {-mrun * c, -mrun * c};
#pragma unroll
        for (int h8 = 0; h8 < 4; ++h8) {
          f32x2 e[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int r = 8 * (h8 & 1) + 2 * k;
            f32x2 a = (h8 < 2) ? f32x2{s0[r], s0[r + 1]} : f32x2{s1[r], s1[r + 1]};
            a = __builtin_elementwise_fma(a, c2, nmc2);
            a.x = __builtin_amdgcn_exp2f(a.x);
            a.y = __builtin_amdgcn_exp2f(a.y);
            l2 += a;
            e[k] = a;
          }
          pb[h8] = h16_pack8<DT>(e[0].x, e[0].y, e[1].x, e[1].y, e[2].x, e[2].y, e[3].x, e[3].y);
        }
      }
    }
    if ((p & 1) == 1 && j < ntile) {
      *reinterpret_cast<uint4*>(Ks + ((j + 1) & 1) * (HKV * HLD) + soff) = rk;
      *reinterpret_cast<uint4*>(Vs + (j & 1) * (HKV * HLD) + soff) = rv;
    }
    __syncthreads();
  }
  if (!wave_active) return;
  const int q = qw0 + l31;
  const float inv = 1.0f / h_xhalf_sum(l2.x + l2.y);
  if (q < len) {
    u16* op = out + (size_t)(seg0 + q) * (heads * 64) + head * 64 + 4 * hi;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *reinterpret_cast<uint2*>(op + 8 * g) =
          h16_pack4<DT>(o0[4 * g + 0] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
      *reinterpret_cast<uint2*>(op + 32 + 8 * g) =
          h16_pack4<DT>(o1[4 * g + 0] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Software-pipelined bounded-softmax kernel (r02; rap_set_tuning(3, 12), NOT the default).  MEASURED (r02 calls 12 / 13): 817 / 922 TF vs
// 866 / 970 for the r01 kernel -- per WAVE it is 1.65x more efficient (SQ_WAVE_CYCLES 7.7e9 vs 12.7e9 for the same two launches), but its
// 194 VGPRs (two score sets + prefetched fragments) allow one 8-wave block per CU where the r01 kernel (122 VGPRs) runs two, and four
// waves per SIMD hide more than the in-wave overlap buys.  The counters say what bounds both: 7.3 VALU + 1 LDS + ~1 scalar instruction
// per MFMA against the ~5 that fit into the shadow of a 32-cycle MFMA (matrix pipe 44 % busy, VALU issue 52 %, LDS 10 %).  Kept as the
// A/B evidence for that statement and as the base for a 64-query-per-wave version (half the fragment reads per MFMA).
// PMC on the r01 kernel: 43 % of the matrix peak with the VALU softmax and the two MFMA groups of a key tile strictly
// one after the other inside a wave -- and both waves of a SIMD in the same phase after every barrier.  Here a wave overlaps them
// itself: while the matrix pipe works on S(i+1) = K(i+1) Q^T and O += V(i)^T P(i), the wave issues the exponentials / conversions /
// row sums of tile i in the shadow of those MFMAs.  One key tile = four QUARTERS, each
//     2 MFMAs of S(i+1) (d-chunk ks)  |  8 v_exp_f32 + 4 v_cvt_pk + adds of P(i) columns 16 ks .. 16 ks + 15  |  2 MFMAs of O += V(i) P(i)
// pinned with sched_barrier(0); the fragment reads of a quarter are issued one quarter ahead.  S lives in two register sets that swap
// roles every tile (the loop body is instantiated twice).  LDS: K(i+1), V(i) being read while K(i+2), V(i+1) are parked for the next
// iteration (both double-buffered, one barrier per tile as before).  Bounded softmax only: p = exp2(score) (PRE: q pre-scaled by
// log2(e)/8, no offset) or p = exp2((s - 8 B) c); no running maximum, no rescale, so nothing in the tile depends on the previous one
// except O and the row sum.
// ---------------------------------------------------------------------------------------------
template <int DT, int PRE>
__global__ __launch_bounds__(512, 2) void attention_h16_sp_kernel(const u16* __restrict__ qk, const u16* __restrict__ vt,
                                                                  int vt_nblk, u16* __restrict__ out, int TP, int heads,
                                                                  const AttnWorkItem* __restrict__ items,
                                                                  const float* __restrict__ bound) {
  typedef typename H16<DT>::T8 T8;
  __shared__ __attribute__((aligned(16))) u16 smem[4 * HKV * HLD];
  u16* Ks = smem;                    // [2][64 keys][72]
  u16* Vs = smem + 2 * HKV * HLD;    // [2][64 d][72]

  const int head = blockIdx.x % heads;
  const AttnWorkItem it = items[blockIdx.x / heads];
  const int len = it.seg_len;
  if (len <= 0) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int seg0 = it.seg_start, seg1 = it.seg_start + len;

  const u16* Qg = qk + (size_t)head * TP * 64;
  const u16* Kg = qk + (size_t)(heads + head) * TP * 64;
  const u16* Vg = vt + (size_t)head * vt_nblk * (64 * 64);

  const int qw0 = it.q0 + wave * 32;
  const bool wave_active = qw0 < len;

  T8 qf[4];
  {
    int q = qw0 + l31;
    q = q < len ? q : len - 1;
    const u16* qp = Qg + (size_t)(seg0 + q) * 64 + 8 * hi;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(qp + 16 * s));
  }
  f32x16 o0, o1, sa0, sa1, sb0, sb1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; sa0[r] = 0.f; sa1[r] = 0.f; sb0[r] = 0.f; sb1[r] = 0.f; }
  const float c = 0.125f * 1.44269504088896340736f;
  const float nmc = PRE ? 0.f : -bound[head] * 8.0f * c;
  f32x2 psq[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

  const int srow = tid >> 3, sch = (tid & 7) * 8;
  const int b_first = seg0 >> 6;
  const int ntile = ((seg1 - 1) >> 6) - b_first + 1;
  const int soff = srow * HLD + sch;
  uint4 rk, rv;
  // prologue: K(0), K(1), V(0) into the LDS
  {
    int tok = b_first * 64 + srow; tok = tok < TP ? tok : TP - 1;
    *reinterpret_cast<uint4*>(Ks + soff) = *reinterpret_cast<const uint4*>(Kg + (size_t)tok * 64 + sch);
    tok = (b_first + 1) * 64 + srow; tok = tok < TP ? tok : TP - 1;
    *reinterpret_cast<uint4*>(Ks + HKV * HLD + soff) = *reinterpret_cast<const uint4*>(Kg + (size_t)tok * 64 + sch);
    *reinterpret_cast<uint4*>(Vs + soff) = *reinterpret_cast<const uint4*>(Vg + ((size_t)b_first * 64 + srow) * 64 + sch);
  }
  __syncthreads();
  const int lrow = l31 * HLD + 8 * hi;
  if (wave_active) {                          // S(0) = K(0) Q^T (not overlapped: once per block)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const T8 k0 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(Ks + lrow + 16 * s));
      const T8 k1 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(Ks + lrow + 32 * HLD + 16 * s));
      sa0 = H16<DT>::mfma(k0, qf[s], sa0);
      sa1 = H16<DT>::mfma(k1, qf[s], sa1);
    }
  }
#define SP_SB __builtin_amdgcn_sched_barrier(0);
  // exponentials of 8 scores (registers RB .. RB+7 of SC) -> the 16-bit B operand of one P*V step; row sum into ps2
#define SP_EXP8(SC, RB, PB, PS)                                                                          \
  {                                                                                                  \
    float e_[8];                                                                                     \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                  \
      const float a_ = PRE ? SC[(RB) + u] : __builtin_fmaf(SC[(RB) + u], c, nmc);                    \
      e_[u] = __builtin_amdgcn_exp2f(a_);                                                            \
    }                                                                                                \
    PB = h16_pack8<DT>(e_[0], e_[1], e_[2], e_[3], e_[4], e_[5], e_[6], e_[7]);                      \
    PS += (f32x2{e_[0], e_[1]} + f32x2{e_[2], e_[3]}) + (f32x2{e_[4], e_[5]} + f32x2{e_[6], e_[7]});   /* one accumulator per quarter: no chain */ \
  }
  // one key tile: SC = scores of tile T (complete), SN = accumulators of tile T+1 (zeroed here, complete afterwards)
#define SP_TILE(T, SC0, SC1, SN0, SN1)                                                               \
  {                                                                                                  \
    const int t_ = (T);                                                                              \
    const bool more_ = t_ + 1 < ntile;                    /* tile t+1 exists: its scores are computed here */                  \
    const bool stage_ = t_ + 2 < ntile;                   /* K(t+2) */                                                        \
    if (more_) {                                                                                     \
      const int blk_ = b_first + t_ + 1;                                                             \
      rv = *reinterpret_cast<const uint4*>(Vg + ((size_t)blk_ * 64 + srow) * 64 + sch);              \
      int tok_ = (blk_ + 1) * 64 + srow; tok_ = tok_ < TP ? tok_ : TP - 1;                           \
      rk = *reinterpret_cast<const uint4*>(Kg + (size_t)tok_ * 64 + sch);                            \
    }                                                                                                \
    if (wave_active) {                                                                               \
      const int tile0_ = (b_first + t_) * 64;                                                        \
      if (tile0_ < seg0 || tile0_ + 64 > seg1) {                                                     \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
          const int kg_ = tile0_ + mfma32_crow(r, hi);                                               \
          SC0[r] = (kg_ >= seg0 && kg_ < seg1) ? SC0[r] : -1e30f;                                    \
          SC1[r] = (kg_ + 32 >= seg0 && kg_ + 32 < seg1) ? SC1[r] : -1e30f;                          \
        }                                                                                            \
      }                                                                                              \
      const u16* kp_ = Ks + ((t_ + 1) & 1) * (HKV * HLD) + lrow;                                     \
      const u16* vp_ = Vs + (t_ & 1) * (HKV * HLD) + lrow;                                           \
      uint4 fk0_ = *reinterpret_cast<const uint4*>(kp_), fk1_ = *reinterpret_cast<const uint4*>(kp_ + 32 * HLD);               \
      uint4 fv0_ = *reinterpret_cast<const uint4*>(vp_), fv1_ = *reinterpret_cast<const uint4*>(vp_ + 32 * HLD);               \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                             \
        uint4 nk0_ = fk0_, nk1_ = fk1_, nv0_ = fv0_, nv1_ = fv1_;                                    \
        if (ks < 3) {                                     /* fragments of the next quarter */                                  \
          nk0_ = *reinterpret_cast<const uint4*>(kp_ + 16 * (ks + 1)); nk1_ = *reinterpret_cast<const uint4*>(kp_ + 32 * HLD + 16 * (ks + 1)); \
          nv0_ = *reinterpret_cast<const uint4*>(vp_ + 16 * (ks + 1)); nv1_ = *reinterpret_cast<const uint4*>(vp_ + 32 * HLD + 16 * (ks + 1)); \
        }                                                                                            \
        SP_SB                                                                                        \
        if (more_) {                                                                                 \
          SN0 = H16<DT>::mfma(__builtin_bit_cast(T8, fk0_), qf[ks], ks == 0 ? zero16 : SN0);        /* C = 0: an inline constant, no v_mov */ \
          SN1 = H16<DT>::mfma(__builtin_bit_cast(T8, fk1_), qf[ks], ks == 0 ? zero16 : SN1);         \
        }                                                                                            \
        SP_SB                                                                                        \
        T8 pb_;                                                                                      \
        if ((ks >> 1) == 0) SP_EXP8(SC0, 8 * (ks & 1), pb_, psq[ks]) else SP_EXP8(SC1, 8 * (ks & 1), pb_, psq[ks])     \
        SP_SB                                                                                        \
        o0 = H16<DT>::mfma(__builtin_bit_cast(T8, fv0_), pb_, o0);                                   \
        o1 = H16<DT>::mfma(__builtin_bit_cast(T8, fv1_), pb_, o1);                                   \
        SP_SB                                                                                        \
        fk0_ = nk0_; fk1_ = nk1_; fv0_ = nv0_; fv1_ = nv1_;                                          \
      }                                                                                              \
    }                                                                                                \
    if (more_) {                                                                                     \
      *reinterpret_cast<uint4*>(Vs + ((t_ + 1) & 1) * (HKV * HLD) + soff) = rv;                      \
      if (stage_) *reinterpret_cast<uint4*>(Ks + (t_ & 1) * (HKV * HLD) + soff) = rk;                \
    }                                                                                                \
    __syncthreads();                                                                                 \
  }

  for (int t = 0; t < ntile; t += 2) {
    SP_TILE(t, sa0, sa1, sb0, sb1)
    if (t + 1 < ntile) SP_TILE(t + 1, sb0, sb1, sa0, sa1)
  }

  if (!wave_active) return;
  const int q = qw0 + l31;
  const f32x2 pst = (psq[0] + psq[1]) + (psq[2] + psq[3]);
  const float inv = 1.0f / h_xhalf_sum(pst.x + pst.y);
  if (q < len) {
    u16* op = out + (size_t)(seg0 + q) * (heads * 64) + head * 64 + 4 * hi;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *reinterpret_cast<uint2*>(op + 8 * g) =
          h16_pack4<DT>(o0[4 * g + 0] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
      *reinterpret_cast<uint2*>(op + 32 + 8 * g) =
          h16_pack4<DT>(o1[4 * g + 0] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Software-pipelined kernel, second form (r02 calls 24-26; rap_set_tuning(3, 13)).  scripts/attn_mix.hip runs the instruction mix of
// this loop with no global traffic in six arrangements (profiles/r02_c24..c26_attn_mix*.jsonl):
//     phase-serial + barrier per tile (the r01 kernel's structure)            1 200-1 250 TF
//     pipelined across tiles, groups pinned with sched_barrier (the kernel above)  1 200-1 300
//     pipelined, order left to the compiler                                    1 490-1 560
//     ... + one block barrier per TILE                                         1 120-1 220   <- the barrier splits the scheduling region
//     ... + one block barrier per TWO tiles                                    1 410-1 450
// so this kernel drops the pins, keeps two tiles in one basic block (the two score register sets swap roles inside it) and
// synchronises the block once per two tiles: K / V^T are staged TWO tiles ahead into a two-stage ring of two-tile stages
// (K stage j = K tiles 2j+1, 2j+2; V stage j = V tiles 2j, 2j+1; K(0) has its own buffer), 83 KB of LDS, one 8-wave block per CU.
// Row sums are plain v_add_f32 on eight independent accumulators (packed fp32 beside MFMAs measured 3 % slower in the mix).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sp2_fadd(float a, float b) {   // one v_add_f32 the SLP vectoriser cannot turn into v_pk_add_f32.
  float r = a + b;              // NOT an inline-asm add: hipcc's hazard recogniser cannot see into asm, and a VALU read of a
  asm("" : "+v"(r));            // v_exp_f32 result needs a wait state (r02 call 27: row sums of stale registers); the empty asm only
  return r;                     // makes the value opaque to the vectoriser and emits nothing.
}
#define SP2_TILE_U16 (HKV * HLD)
#define SP2_LDS_BYTES (9 * SP2_TILE_U16 * 2)
// ABL (timing only, RAP_ABLATION_BUILD): 1 = no K / V^T stream (every tile re-reads stage 0), 2 = additionally no barriers
template <int DT, int PRE, int ABL = 0>
__global__ __launch_bounds__(512, 2) void attention_h16_sp2_kernel(const u16* __restrict__ qk, const u16* __restrict__ vt,
                                                                   int vt_nblk, u16* __restrict__ out, int TP, int heads,
                                                                   const AttnWorkItem* __restrict__ items,
                                                                   const float* __restrict__ bound) {
  typedef typename H16<DT>::T8 T8;
  extern __shared__ __attribute__((aligned(16))) u16 sp2_smem[];
  u16* K0s = sp2_smem;                              // K tile 0
  u16* Kst = sp2_smem + SP2_TILE_U16;               // [stage 2][slot 2][64 keys][72]
  u16* Vst = sp2_smem + 5 * SP2_TILE_U16;           // [stage 2][slot 2][64 d][72]

  const int head = blockIdx.x % heads;
  const AttnWorkItem it = items[blockIdx.x / heads];
  const int len = it.seg_len;
  if (len <= 0) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int seg0 = it.seg_start, seg1 = it.seg_start + len;

  const u16* Qg = qk + (size_t)head * TP * 64;
  const u16* Kg = qk + (size_t)(heads + head) * TP * 64;
  const u16* Vg = vt + (size_t)head * vt_nblk * (64 * 64);

  const int qw0 = it.q0 + wave * 32;
  const bool wave_active = qw0 < len;

  T8 qf[4];
  {
    int q = qw0 + l31;
    q = q < len ? q : len - 1;
    const u16* qp = Qg + (size_t)(seg0 + q) * 64 + 8 * hi;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(qp + 16 * s));
  }
  f32x16 o0, o1, sa0, sa1, sb0, sb1, zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; sa0[r] = 0.f; sa1[r] = 0.f; sb0[r] = 0.f; sb1[r] = 0.f; zero16[r] = 0.f; }
  const float c = 0.125f * 1.44269504088896340736f;
  const float nmc = PRE ? 0.f : -bound[head] * 8.0f * c;
  float ps[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const int srow = tid >> 3, sch = (tid & 7) * 8;
  const int b_first = seg0 >> 6;
  const int ntile = ((seg1 - 1) >> 6) - b_first + 1;
  const int soff = srow * HLD + sch;
  // K tile T / V tile T of this segment (T < ntile); the K row index is clamped, the V^T image is padded to whole blocks
#define SP2_LDK(T) ({ int tok_ = (b_first + (T)) * 64 + srow; tok_ = tok_ < TP ? tok_ : TP - 1;                \
                      *reinterpret_cast<const uint4*>(Kg + (size_t)tok_ * 64 + sch); })
#define SP2_LDV(T) (*reinterpret_cast<const uint4*>(Vg + ((size_t)(b_first + (T)) * 64 + srow) * 64 + sch))
  {  // prologue: K(0); stage 0 = K(1), K(2), V(0), V(1)
    const uint4 k0 = SP2_LDK(0);
    const uint4 v0 = SP2_LDV(0);
    uint4 k1 = k0, k2 = k0, v1 = v0;
    if (1 < ntile) { k1 = SP2_LDK(1); v1 = SP2_LDV(1); }
    if (2 < ntile) k2 = SP2_LDK(2);
    *reinterpret_cast<uint4*>(K0s + soff) = k0;
    *reinterpret_cast<uint4*>(Vst + soff) = v0;
    *reinterpret_cast<uint4*>(Kst + soff) = k1;
    *reinterpret_cast<uint4*>(Vst + SP2_TILE_U16 + soff) = v1;
    *reinterpret_cast<uint4*>(Kst + SP2_TILE_U16 + soff) = k2;
  }
  __syncthreads();
  const int lrow = l31 * HLD + 8 * hi;
  if (wave_active) {                          // S(0) = K(0) Q^T (once per block)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const T8 k0 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(K0s + lrow + 16 * s));
      const T8 k1 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(K0s + lrow + 32 * HLD + 16 * s));
      sa0 = H16<DT>::mfma(k0, qf[s], s == 0 ? zero16 : sa0);
      sa1 = H16<DT>::mfma(k1, qf[s], s == 0 ? zero16 : sa1);
    }
  }
  // the exponentials of 8 scores -> the 16-bit B operand of one P*V step; row sums into ps[0..7]
#define SP2_EXP8(SC, RB, PB)                                                                         \
  {                                                                                                  \
    float e_[8];                                                                                     \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                  \
      const float a_ = PRE ? SC[(RB) + u] : __builtin_fmaf(SC[(RB) + u], c, nmc);                    \
      e_[u] = __builtin_amdgcn_exp2f(a_);                                                            \
      ps[u] = sp2_fadd(ps[u], e_[u]);                                                                \
    }                                                                                                \
    PB = h16_pack8<DT>(e_[0], e_[1], e_[2], e_[3], e_[4], e_[5], e_[6], e_[7]);                      \
  }
  // Tile T: SC = finished (and masked) scores of tile T, SN = accumulators of tile T+1; SLOT = T & 1, stage = (T >> 1) & 1.
  // Straight-line on purpose (one basic block per pair of tiles): the loads of V(T+2) / K(T+3) are unconditional with the tile
  // index clamped to the last tile (a duplicate lands in a slot nobody reads), and S(T+1) is computed even when T+1 does not
  // exist except in the peeled last tile (LAST).
#define SP2_STAGE_LOADS(T, RV, RK)                                                                   \
  {                                                                                                  \
    const int tv_ = (T) + 2 < last ? (T) + 2 : last, tk_ = (T) + 3 < last ? (T) + 3 : last;          \
    RV = SP2_LDV(tv_);                                                                               \
    RK = SP2_LDK(tk_);                                                                               \
  }
#define SP2_STAGE_PARK(T, SLOT, RV, RK)                                                              \
  {                                                                                                  \
    const int so_ = (((((T) >> 1) & 1) ^ 1) * 2 + (SLOT)) * SP2_TILE_U16 + soff;                     \
    *reinterpret_cast<uint4*>(Vst + so_) = RV;                                                       \
    *reinterpret_cast<uint4*>(Kst + so_) = RK;                                                       \
  }
#define SP2_TILE(T, SLOT, SC0, SC1, SN0, SN1, LAST)                                                  \
  {                                                                                                  \
    const int fo_ = ((ABL ? 0 : (((T) >> 1) & 1)) * 2 + (SLOT)) * SP2_TILE_U16 + lrow;               \
    const u16* kp_ = Kst + fo_;                                                                      \
    const u16* vp_ = Vst + fo_;                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                               \
      if (!(LAST)) {                                                                                 \
        const T8 fk0_ = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(kp_ + 16 * ks));      \
        const T8 fk1_ = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(kp_ + 32 * HLD + 16 * ks)); \
        SN0 = H16<DT>::mfma(fk0_, qf[ks], ks == 0 ? zero16 : SN0);                                   \
        SN1 = H16<DT>::mfma(fk1_, qf[ks], ks == 0 ? zero16 : SN1);                                   \
      }                                                                                              \
      T8 pb_;                                                                                        \
      if ((ks >> 1) == 0) SP2_EXP8(SC0, 8 * (ks & 1), pb_) else SP2_EXP8(SC1, 8 * (ks & 1), pb_)     \
      const T8 fv0_ = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(vp_ + 16 * ks));        \
      const T8 fv1_ = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(vp_ + 32 * HLD + 16 * ks)); \
      o0 = H16<DT>::mfma(fv0_, pb_, o0);                                                             \
      o1 = H16<DT>::mfma(fv1_, pb_, o1);                                                             \
    }                                                                                                \
  }
  // keys of tile T outside the segment (only the first and the last tile can have any)
#define SP2_MASK(T, SC0, SC1)                                                                        \
  {                                                                                                  \
    const int tile0_ = (b_first + (T)) * 64;                                                         \
    if (tile0_ < seg0 || tile0_ + 64 > seg1) {                                                       \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                               \
        const int kg_ = tile0_ + mfma32_crow(r, hi);                                                 \
        SC0[r] = (kg_ >= seg0 && kg_ < seg1) ? SC0[r] : -1e30f;                                      \
        SC1[r] = (kg_ + 32 >= seg0 && kg_ + 32 < seg1) ? SC1[r] : -1e30f;                            \
      }                                                                                              \
    }                                                                                                \
  }

  uint4 rk0, rv0, rk1, rv1;
  const int last = ntile - 1;
  int t = 0;
  if (!wave_active) {                         // a wave without queries only stages its share of K / V^T (same barriers as the others)
    for (; t + 2 <= last; t += 2) {
      SP2_STAGE_LOADS(t, rv0, rk0) SP2_STAGE_LOADS(t + 1, rv1, rk1)
      SP2_STAGE_PARK(t, 0, rv0, rk0) SP2_STAGE_PARK(t + 1, 1, rv1, rk1)
      __syncthreads();
    }
    return;
  }
  SP2_MASK(0, sa0, sa1)
  for (; t + 2 <= last; t += 2) {             // tiles t, t + 1 (neither is the last one): one basic block, one barrier
    if (!ABL) { SP2_STAGE_LOADS(t, rv0, rk0) SP2_STAGE_LOADS(t + 1, rv1, rk1) }   // the next stage: issued first, parked last (hipcc would sink them to the stores)
    __builtin_amdgcn_sched_barrier(0);
    SP2_TILE(t, 0, sa0, sa1, sb0, sb1, false)
    SP2_TILE(t + 1, 1, sb0, sb1, sa0, sa1, false)
    __builtin_amdgcn_sched_barrier(0);
    if (!ABL) { SP2_STAGE_PARK(t, 0, rv0, rk0) SP2_STAGE_PARK(t + 1, 1, rv1, rk1) }
    if (ABL < 2) __syncthreads();
  }
  if (t < last) {                             // last is odd: tile last - 1 (slot 0), then the masked last tile (slot 1) from the same stage
    SP2_TILE(t, 0, sa0, sa1, sb0, sb1, false)
    if (last > 0) SP2_MASK(last, sb0, sb1)
    SP2_TILE(t + 1, 1, sb0, sb1, sa0, sa1, true)
  } else {                                    // last is even: the masked last tile (slot 0)
    if (last > 0) SP2_MASK(last, sa0, sa1)
    SP2_TILE(t, 0, sa0, sa1, sb0, sb1, true)
  }

  const int q = qw0 + l31;
  const float inv = 1.0f / h_xhalf_sum(((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7])));
  if (q < len) {
    u16* op = out + (size_t)(seg0 + q) * (heads * 64) + head * 64 + 4 * hi;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *reinterpret_cast<uint2*>(op + 8 * g) =
          h16_pack4<DT>(o0[4 * g + 0] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
      *reinterpret_cast<uint2*>(op + 32 + 8 * g) =
          h16_pack4<DT>(o1[4 * g + 0] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
    }
  }
}

// tuning knob (rap_set_tuning key 3): 0 = v1 (bounded softmax when per-head logit bounds are supplied -- bf16 only -- else
// v_max3 row maximum + deferred rescale); 5 = v1 online softmax even with bounds; 11 = ping-pong schedule (slower, see above); 8 = first v1 (fmaxf chain, rescale
// every tile); 4 = max3 only; 1..3, 6, 7 = timing-only ablations (bf16 only), see ABL above.
// 9 = as 0 but the model path keeps q un-scaled (the per-score FMA form, for A/B timing of the pre-scaled default).
rap_tuning_t g_rap_attn_h16_variant = 0;

// work-list granularity of the selected schedule (variant 24: 512-query blocks, bf16 only)
int attention_h16_block_queries(int dtype) { return (g_rap_attn_h16_variant == 24 && dtype == RAP_DT_BF16) ? 512 : 256; }

// the model path asks before it runs qk-norm: pre-scaled q only feeds the default bounded bf16 kernel
bool attention_h16_wants_prescaled_q(int dtype, bool bounded) {
  const int v = g_rap_attn_h16_variant;
  return dtype == RAP_DT_BF16 && bounded && (v == 0 || v == 9 || v == 10 || v == 12 || v == 13 || v == 19 || v == 20 || v == 22 || v == 23 || v == 24);
}

int launch_attention_h16(hipStream_t stream, int dtype, const u16* qk, const u16* vt, int vt_nblk, u16* out, int TP,
                         int heads, const AttnWorkItem* items, int max_items, const float* bound, int q_prescaled) {
  if (max_items <= 0 || TP <= 0) return RAP_OK;
  if (heads <= 0 || vt_nblk * 64 < TP) return RAP_ERR_INVALID;
  if (q_prescaled && !(bound && attention_h16_wants_prescaled_q(dtype, true))) return RAP_ERR_INVALID;
#define HPP_LAUNCH(DTV, BV) \
  hipLaunchKernelGGL((attention_h16_pp_kernel<DTV, BV>), dim3(max_items * heads), dim3(512), 0, stream, qk, vt, vt_nblk, out, TP, heads, items, bound)
  if (g_rap_attn_h16_variant == 11) {
    if (dtype == RAP_DT_BF16) { if (bound) HPP_LAUNCH(RAP_DT_BF16, true); else HPP_LAUNCH(RAP_DT_BF16, false); }
    else if (dtype == RAP_DT_F16) HPP_LAUNCH(RAP_DT_F16, false);
    else return RAP_ERR_INVALID;
    RAP_LAUNCH_CHECK();
    return RAP_OK;
  }
#define HATT_LAUNCH(DTV, ABLV, OPTV) \
  hipLaunchKernelGGL((attention_h16_kernel<DTV, ABLV, OPTV>), dim3(max_items * heads), dim3(512), 0, stream, qk, vt, vt_nblk, out, TP, heads, items, bound, max_items * heads)
  // the shipped default since r02 call 54: the same kernel with the output stored as whole rows through an LDS slab (LST): +0.2 ... 1.8 % per
  // launch, bench 106.8 k -> 107.4 k points/s; rap_set_tuning(3, 23) = the direct 8-byte stores
#define HATT_LAUNCH_D(DTV, OPTV) \
  hipLaunchKernelGGL((attention_h16_kernel<DTV, 0, OPTV, false, false, true>), dim3(max_items * heads), dim3(512), 0, stream, qk, vt, vt_nblk, out, TP, heads, items, bound, max_items * heads)
  if (dtype == RAP_DT_BF16) {
    switch (g_rap_attn_h16_variant) {
#ifdef RAP_ABLATION_BUILD      // timing-only kernels whose output is NOT attention: never part of the shipped library
      case 1: HATT_LAUNCH(RAP_DT_BF16, 1, 0); break;
      case 2: HATT_LAUNCH(RAP_DT_BF16, 2, 0); break;
      case 3: HATT_LAUNCH(RAP_DT_BF16, 3, 0); break;
      case 6: HATT_LAUNCH(RAP_DT_BF16, 8, 3); break;     // ... without transcendentals
      case 7: HATT_LAUNCH(RAP_DT_BF16, 32, 3); break;    // ... without the max chain
      case 21: if (bound) HATT_LAUNCH(RAP_DT_BF16, 256, 8); break;   // bounded kernel with s_memtime stamps (rap_debug_attn_ts)
      case 16: if (bound) HATT_LAUNCH(RAP_DT_BF16, 64, 8); break;    // bounded kernel without the output stores
      case 17: if (bound) HATT_LAUNCH(RAP_DT_BF16, 128, 8); break;   // ... without the Q loads
      case 18: if (bound) HATT_LAUNCH(RAP_DT_BF16, 192, 8); break;   // ... without either
#endif
      case 4: HATT_LAUNCH(RAP_DT_BF16, 0, 7); break;     // + s_setprio(1) around the MFMA clusters
      case 5: HATT_LAUNCH(RAP_DT_BF16, 0, 3); break;     // max3 + deferred rescale
      case 8: HATT_LAUNCH(RAP_DT_BF16, 0, 0); break;     // v1: fmaxf chain, rescale every tile
      case 12:                                            // software-pipelined bounded kernel (r02: measured 5 % slower, see its header)
        if (bound && q_prescaled)
          hipLaunchKernelGGL((attention_h16_sp_kernel<RAP_DT_BF16, 1>), dim3(max_items * heads), dim3(512), 0, stream, qk, vt, vt_nblk, out, TP, heads, items, bound);
        else if (bound)
          hipLaunchKernelGGL((attention_h16_sp_kernel<RAP_DT_BF16, 0>), dim3(max_items * heads), dim3(512), 0, stream, qk, vt, vt_nblk, out, TP, heads, items, bound);
        else HATT_LAUNCH(RAP_DT_BF16, 0, 3);
        break;
#ifdef RAP_ABLATION_BUILD
      case 14: case 15: {
        const void* f = g_rap_attn_h16_variant == 14 ? reinterpret_cast<const void*>(&attention_h16_sp2_kernel<RAP_DT_BF16, 0, 1>)
                                                     : reinterpret_cast<const void*>(&attention_h16_sp2_kernel<RAP_DT_BF16, 0, 2>);
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, SP2_LDS_BYTES) != hipSuccess || !bound) return RAP_ERR_HIP;
        if (g_rap_attn_h16_variant == 14)
          hipLaunchKernelGGL((attention_h16_sp2_kernel<RAP_DT_BF16, 0, 1>), dim3(max_items * heads), dim3(512), SP2_LDS_BYTES, stream, qk, vt, vt_nblk, out, TP, heads, items, bound);
        else
          hipLaunchKernelGGL((attention_h16_sp2_kernel<RAP_DT_BF16, 0, 2>), dim3(max_items * heads), dim3(512), SP2_LDS_BYTES, stream, qk, vt, vt_nblk, out, TP, heads, items, bound);
        break;
      }
#endif
      case 13: {                                          // software-pipelined, compiler-scheduled, one barrier per two tiles
        static bool sp2_attr = false;
        if (!sp2_attr) {
          if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_h16_sp2_kernel<RAP_DT_BF16, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, SP2_LDS_BYTES) != hipSuccess ||
              hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_h16_sp2_kernel<RAP_DT_BF16, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, SP2_LDS_BYTES) != hipSuccess)
            return RAP_ERR_HIP;
          sp2_attr = true;
        }
        if (bound && q_prescaled)
          hipLaunchKernelGGL((attention_h16_sp2_kernel<RAP_DT_BF16, 1>), dim3(max_items * heads), dim3(512), SP2_LDS_BYTES, stream, qk, vt, vt_nblk, out, TP, heads, items, bound);
        else if (bound)
          hipLaunchKernelGGL((attention_h16_sp2_kernel<RAP_DT_BF16, 0>), dim3(max_items * heads), dim3(512), SP2_LDS_BYTES, stream, qk, vt, vt_nblk, out, TP, heads, items, bound);
        else HATT_LAUNCH(RAP_DT_BF16, 0, 3);
        break;
      }
      case 24: {                                          // 512-query blocks, 16 waves (the caller built the work list with 512-query items)
#define HATT_LAUNCH_W(OPTV) \
  hipLaunchKernelGGL((attention_h16_kernel<RAP_DT_BF16, 0, OPTV, false, false, false, true>), dim3(max_items * heads), dim3(1024), 0, stream, qk, vt, vt_nblk, out, TP, heads, items, bound, max_items * heads)
        if (bound && q_prescaled) HATT_LAUNCH_W(24);
        else if (bound) HATT_LAUNCH_W(8);
        else HATT_LAUNCH_W(3);
        break;
      }
      case 22:                                            // output stored as whole rows through an LDS slab
#define HATT_LAUNCH_L(OPTV) \
  hipLaunchKernelGGL((attention_h16_kernel<RAP_DT_BF16, 0, OPTV, false, false, true>), dim3(max_items * heads), dim3(512), 0, stream, qk, vt, vt_nblk, out, TP, heads, items, bound, max_items * heads)
        if (bound && q_prescaled) HATT_LAUNCH_L(24);
        else if (bound) HATT_LAUNCH_L(8);
        else HATT_LAUNCH_L(3);
        break;
      case 20:                                            // rotated key-tile walk
#define HATT_LAUNCH_R(OPTV) \
  hipLaunchKernelGGL((attention_h16_kernel<RAP_DT_BF16, 0, OPTV, false, true>), dim3(max_items * heads), dim3(512), 0, stream, qk, vt, vt_nblk, out, TP, heads, items, bound, max_items * heads)
        if (bound && q_prescaled) HATT_LAUNCH_R(24);
        else if (bound) HATT_LAUNCH_R(8);
        else HATT_LAUNCH_R(3);
        break;
      case 19: {                                          // persistent blocks (two per CU) walking the work list
        const int total = max_items * heads;
        int grid = (512 / heads) * heads;
        if (grid <= 0 || grid > total) grid = total;
#define HATT_LAUNCH_P(OPTV) \
  hipLaunchKernelGGL((attention_h16_kernel<RAP_DT_BF16, 0, OPTV, true>), dim3(grid), dim3(512), 0, stream, qk, vt, vt_nblk, out, TP, heads, items, bound, total)
        if (bound && q_prescaled) HATT_LAUNCH_P(24);
        else if (bound) HATT_LAUNCH_P(8);
        else HATT_LAUNCH_P(3);
        break;
      }
      case 23:                                            // the r01 epilogue: direct 8-byte stores
        if (bound && q_prescaled) HATT_LAUNCH(RAP_DT_BF16, 0, 24);
        else if (bound) HATT_LAUNCH(RAP_DT_BF16, 0, 8);
        else HATT_LAUNCH(RAP_DT_BF16, 0, 3);
        break;
      default:
        if (bound && q_prescaled) HATT_LAUNCH_D(RAP_DT_BF16, 24);
        else if (bound) HATT_LAUNCH_D(RAP_DT_BF16, 8);
        else HATT_LAUNCH_D(RAP_DT_BF16, 3);
        break;
    }
  } else if (dtype == RAP_DT_F16) {
    if (g_rap_attn_h16_variant == 8) HATT_LAUNCH(RAP_DT_F16, 0, 0);
    else if (g_rap_attn_h16_variant == 23) HATT_LAUNCH(RAP_DT_F16, 0, 3);
    else HATT_LAUNCH_D(RAP_DT_F16, 3);
  } else {
    return RAP_ERR_INVALID;
  }
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
