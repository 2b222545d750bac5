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
// DMA (round 3, the default): K / V^T tiles go global -> LDS directly (global_load_lds_dwordx4, two 1 KB pieces per wave and tile: no
// staging registers, no ds_write_b128, no s_waitcnt in front of them -- round 2's instruction-mix microbenchmark priced the
// register-staged stream at 16-20 % of this loop).  The DMA writes LDS lane-linearly, so rows are 128 bytes without padding and
// bank conflicts are removed as in the GEMMs: 16-byte slot' = slot ^ ((row >> 1) & 7), applied to the per-lane global source
// address and to the ds_read_b128 address.
template <int DT, int ABL, int OPT, bool PERSIST = false, bool ROT = false, bool LST = false, bool W16 = false, bool DMA = false>
__global__ __launch_bounds__(W16 ? 1024 : 512, 2) void attention_h16_kernel(const u16* __restrict__ qk, const u16* __restrict__ vt,
                                                               int vt_nblk, u16* __restrict__ out, int TP, int heads,
                                                               const AttnWorkItem* __restrict__ items,
                                                               const float* __restrict__ bound, int total_blocks) {
  typedef typename H16<DT>::T8 T8;
  static_assert(!(DMA && (W16 || ROT)), "the LDS-DMA stream is built for the 8-wave, in-order key walk");
  constexpr int LDR = DMA ? 64 : HLD;   // LDS row stride in 16-bit elements: 128 B swizzled (DMA) or 144 B padded
  // (a ring of THREE stages with tiles requested two ahead and a counted vmcnt(2) was measured too, r03 call 21: 1 132 vs 1 141 TF in the
  // bench -- the DMA latency is not exposed with two.)
  __shared__ __attribute__((aligned(16))) u16 smem[4 * HKV * HLD];   // 36 KB: two stages of K and V^T (32 KB with DMA) / the 8 output slabs
  u16* Ks = smem;                    // [2][64 keys][LDR]
  u16* Vs = smem + 2 * HKV * LDR;    // [2][64 d][LDR]   (columns = vt_pos of the key)

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
  // DMA: wave w stages rows 8w .. 8w+7 of the K tile and of the V^T tile; lane -> (row 8w + lane/8, physical slot lane%8)
  const int drow = (tid >> 6) * 8 + ((tid & 63) >> 3);
  const int dls = ((tid & 7) ^ ((drow >> 1) & 7)) * 8;              // logical slot (in elements) this lane fetches
  const unsigned lds_k = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) u16*)Ks + (unsigned)(tid >> 6) * 1024u);
  const unsigned lds_v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) u16*)Vs + (unsigned)(tid >> 6) * 1024u);
#define HATT_DMA1(GSRC, LDSB)                                                                                 \
  {                                                                                                           \
    unsigned keep_;                                                                                           \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(GSRC), "s"(LDSB) : "memory");                                           \
  }
#define HATT_DMA(T, BUF)                                                                             \
  {                                                                                                  \
    const int blk_ = b_first + (T);                                                                  \
    int tok_ = blk_ * 64 + drow;                                                                     \
    tok_ = tok_ < TP ? tok_ : TP - 1;                                                                \
    HATT_DMA1(Kg + (size_t)tok_ * 64 + dls, lds_k + (unsigned)(BUF) * (HKV * 64 * 2))                \
    HATT_DMA1(Vg + ((size_t)blk_ * 64 + drow) * 64 + dls, lds_v + (unsigned)(BUF) * (HKV * 64 * 2))  \
  }
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
  if (DMA) {
    HATT_DMA(0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the Q fragments have landed too -- tell the compiler (a use of every fragment), or it waits for them with vmcnt(3..0) inside
    // the key loop, where those waits would drain the DMA pieces of the NEXT tile it does not know about
#pragma unroll
    for (int s = 0; s < 4; ++s) { uint4 q_ = __builtin_bit_cast(uint4, qf[s]); asm volatile("" : "+v"(q_.x), "+v"(q_.y), "+v"(q_.z), "+v"(q_.w)); qf[s] = __builtin_bit_cast(T8, q_); }
  } else {
    HATT_LOAD(rt)
    HATT_STORE(0)
  }
  __syncthreads();
  ATT_TS(2)
  if (ABL & 256) { const uint4 q0_ = __builtin_bit_cast(uint4, qf[0]); const uint4 q3_ = __builtin_bit_cast(uint4, qf[3]); if ((q0_.x ^ q3_.w) == 0x9e3779b9u && tid == 9999) return; }   // the Q loads complete before stamp 3
  ATT_TS(3)

  for (int t = 0; t < ntile; ++t) {
    const int cur = (ABL & 2) ? 0 : (t & 1);
    const bool more = (ABL & 2) ? false : (t + 1) < ntile;
    const int rt_cur = rt;
    if (ROT) { rt = rt + 1; rt = rt == ntile ? 0 : rt; }
    if (more) { if (DMA) { HATT_DMA(t + 1, cur ^ 1) } else { HATT_LOAD(ROT ? rt : t + 1) } }   // DMA: every wave left buffer cur^1 at the last barrier

    if (wave_active) {
      // ---- S^T = K Q^T : two 32-key sub-tiles x 32 queries
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
      const int swz = (l31 >> 1) & 7;                 // DMA layout: slot ^ ((row >> 1) & 7), the same for rows l31 and 32 + l31
      const u16* kp = Ks + cur * (HKV * LDR) + l31 * LDR + (DMA ? 0 : 8 * hi);
      if (OPT & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int ko = DMA ? ((2 * s + hi) ^ swz) * 8 : 16 * s;
        const T8 k0 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(kp + ko));
        const T8 k1 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(kp + 32 * LDR + ko));
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
      const u16* vp = Vs + cur * (HKV * LDR) + l31 * LDR + (DMA ? 0 : 8 * hi);
      if (OPT & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int rb = 8 * (ks & 1);
        T8 pb;
        if ((ks >> 1) == 0)
          pb = h16_pack8<DT>(s0[rb + 0], s0[rb + 1], s0[rb + 2], s0[rb + 3], s0[rb + 4], s0[rb + 5], s0[rb + 6], s0[rb + 7]);
        else
          pb = h16_pack8<DT>(s1[rb + 0], s1[rb + 1], s1[rb + 2], s1[rb + 3], s1[rb + 4], s1[rb + 5], s1[rb + 6], s1[rb + 7]);
        const int vo = DMA ? ((2 * ks + hi) ^ swz) * 8 : 16 * ks;
        const T8 v0 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(vp + vo));
        const T8 v1 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(vp + 32 * LDR + vo));
        o0 = H16<DT>::mfma(v0, pb, o0);
        o1 = H16<DT>::mfma(v1, pb, o1);
      }
      if (OPT & 4) __builtin_amdgcn_s_setprio(0);
    }

    if (more && !DMA) { HATT_STORE(cur ^ 1) }
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's two pieces of tile t + 1 have landed
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

// Kernel choice per LAUNCH (round 3): per-head logit bounds supplied (the caller guarantees q.k/8 <= bound[h] <= 40) -> the bounded,
// offset-free softmax (bf16: on pre-scaled q when the model path asks for it); no bounds, or fp16 (whose exponent range cannot
// hold 2^58) -> the online softmax with v_max3 row maxima and deferred rescale.  Both are the same kernel template; the output
// leaves as whole rows through an LDS slab.  The other schedules of rounds 1-2 (ping-pong, two software-pipelined kernels,
// persistent blocks, rotated key walk, 512-query blocks -- all measured equal or slower) are in the history at 72efb73.
// RAP_ABLATION_BUILD only: rap_set_tuning(3, 5) forces the online softmax even with bounds (A/B of the bounded kernel).
rap_tuning_t g_rap_attn_h16_variant = 0;
rap_tuning_t g_rap_attn_h16_dma = 1;      // tuning key 13: K / V^T tiles by LDS-DMA (1, default) or staged through registers (0)

int attention_h16_block_queries(int) { return 256; }

// the model path asks before it runs qk-norm: pre-scaled q only feeds the bounded bf16 kernel
bool attention_h16_wants_prescaled_q(int dtype, bool bounded) {
#ifdef RAP_ABLATION_BUILD
  if (g_rap_attn_h16_variant == 5) return false;
#endif
  return dtype == RAP_DT_BF16 && bounded;
}

int launch_attention_h16(hipStream_t stream, int dtype, const u16* qk, const u16* vt, int vt_nblk, u16* out, int TP,
                         int heads, const AttnWorkItem* items, int max_items, const float* bound, int q_prescaled) {
  if (max_items <= 0 || TP <= 0) return RAP_OK;
  if (heads <= 0 || vt_nblk * 64 < TP) return RAP_ERR_INVALID;
  if (q_prescaled && !(bound && attention_h16_wants_prescaled_q(dtype, true))) return RAP_ERR_INVALID;
#ifdef RAP_ABLATION_BUILD
  if (g_rap_attn_h16_variant == 5) bound = nullptr;
#endif
#define HATT_LAUNCH_D(DTV, OPTV) \
  if (g_rap_attn_h16_dma) hipLaunchKernelGGL((attention_h16_kernel<DTV, 0, OPTV, false, false, true, false, true>), dim3(max_items * heads), dim3(512), 0, stream, qk, vt, vt_nblk, out, TP, heads, items, bound, max_items * heads); \
  else hipLaunchKernelGGL((attention_h16_kernel<DTV, 0, OPTV, false, false, true>), dim3(max_items * heads), dim3(512), 0, stream, qk, vt, vt_nblk, out, TP, heads, items, bound, max_items * heads)
  if (dtype == RAP_DT_BF16) {
    if (bound && q_prescaled) { HATT_LAUNCH_D(RAP_DT_BF16, 24); }
    else if (bound) { HATT_LAUNCH_D(RAP_DT_BF16, 8); }
    else { HATT_LAUNCH_D(RAP_DT_BF16, 3); }
  } else if (dtype == RAP_DT_F16) {
    { HATT_LAUNCH_D(RAP_DT_F16, 3); }
  } else {
    return RAP_ERR_INVALID;
  }
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
