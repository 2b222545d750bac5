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
//    double-buffered LDS (LDS-DMA since round 3: the two 1 KB pieces a wave owns of tile t+1 are requested before the MFMAs of
//    tile t and confirmed -- vmcnt(0) -- in front of the one barrier per tile).  Key tiles are the GLOBAL 64-token blocks that intersect the
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
//  * LDS rows are 128 bytes with the 16-byte slot index XOR-swizzled by the row (DMA) or 144 bytes padded (register staging):
//    the 16 rows of each ds_read_b128 lane group hit 16 distinct slots either way.
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

// OPT: bit 0 = row maximum through v_max3_f32, bit 1 = deferred rescale (threshold DEFER_THR in the base-2 exponent) -- together the
// ONLINE softmax; bit 3 = bounded softmax, bit 4 (with bit 3, bf16) = q arrives pre-scaled and the offset is dropped (see below).
// (The timing-only ablation switches, the persistent / rotated / 512-query forms and the in-kernel time stamps of rounds 1-2 are in
// the history at 72efb73; their results are in docs/DESIGN_r01_r02.md section 4.4.)
#define DEFER_THR 11.5f   // = 8 in natural-log units: P <= e^8
__device__ __forceinline__ float hmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// DMA (round 3, the default): K / V^T tiles go global -> LDS directly (global_load_lds_dwordx4, two 1 KB pieces per wave and tile: no
// staging registers, no ds_write_b128, no s_waitcnt in front of them -- round 2's instruction-mix microbenchmark priced the
// register-staged stream at 16-20 % of this loop).  The DMA writes LDS lane-linearly, so rows are 128 bytes without padding and
// bank conflicts are removed as in the GEMMs: 16-byte slot' = slot ^ ((row >> 1) & 7), applied to the per-lane global source
// address and to the ds_read_b128 address.  DMA = false keeps the register-staged stream with 144-byte padded rows (rap_set_tuning(13, 0)).
// The output tile leaves through the wave's own 4.6 KB slab of the (by then idle) K / V^T buffers and is stored as whole 128-byte rows.
// NS / bq (round 6, few-token calls): a demo pair is 64 work items x heads of 16-32 key tiles on 256 CUs -- one block per CU, nothing to hide
// the L2 round trip of the next tile behind (r06: ~1 us per tile whatever the block computes; the r03 sub-block experiment ran into the same
// wall).  NS = 4: a ring of four K / V^T stages, tiles requested three ahead, counted vmcnt.  bq = query rows per work item (64 / 128
// instead of 256: the waves whose rows lie beyond bq only stage -- every wave still issues two pieces per tile): 4 / 2 times the blocks,
// each with a quarter / half of the MFMA work per tile.  Same per-query arithmetic in the same order: bit-identical results.
template <int DT, int OPT, bool DMA, int NS = 2>
__global__ __launch_bounds__(512, 2) void attention_h16_kernel(const u16* __restrict__ qk, const u16* __restrict__ vt, int vt_nblk,
                                                               u16* __restrict__ out, int TP, int heads,
                                                               const AttnWorkItem* __restrict__ items, const float* __restrict__ bound, int bq) {
  typedef typename H16<DT>::T8 T8;
  constexpr int LDR = DMA ? 64 : HLD;   // LDS row stride in 16-bit elements: 128 B swizzled (DMA) or 144 B padded
  static_assert(NS == 2 || (DMA && NS == 4), "two stages, or the four-stage LDS-DMA ring");
  // (at the full configuration -- two blocks per CU cover each other -- a ring of THREE stages was measured in r03 call 21: 1 132 vs
  // 1 141 TF in the bench; there the DMA latency is not exposed with two.)
  constexpr int SMEM_ELEMS = NS == 2 ? 4 * HKV * HLD : NS * 2 * HKV * 64;      // 36 KB (two stages / the 8 output slabs) or 64 KB (ring)
  static_assert(8 * 32 * HLD <= SMEM_ELEMS, "output slabs must fit");
  __shared__ __attribute__((aligned(16))) u16 smem[SMEM_ELEMS];
  u16* Ks = smem;                    // [NS][64 keys][LDR]
  u16* Vs = smem + NS * HKV * LDR;   // [NS][64 d][LDR]   (columns = vt_pos of the key)

  const int tid = threadIdx.x;
  const int head = blockIdx.x % heads;
  const AttnWorkItem it = items[blockIdx.x / heads];
  const int len = it.seg_len;
  if (len <= 0) return;
  const int lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int seg0 = it.seg_start, seg1 = it.seg_start + len;

  const u16* Qg = qk + (size_t)head * TP * 64;
  const u16* Kg = qk + (size_t)(heads + head) * TP * 64;
  const u16* Vg = vt + (size_t)head * vt_nblk * (64 * 64);

  const int qw0 = it.q0 + wave * 32;
  const bool wave_active = wave * 32 < bq && qw0 < len;   // waves beyond the work item's rows / the segment still help stage K/V

  // ---- Q fragments (B operand of S^T): qf[s] = Q[q][16s + 8hi .. +7]
  T8 qf[4];
  {
    int q = qw0 + l31;
    q = q < len ? q : len - 1;
    const u16* qp = Qg + (size_t)(seg0 + q) * 64 + 8 * hi;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(qp + 16 * s));
  }

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  // OPT bit 3: bounded softmax -- the caller guarantees q.k/8 <= bound[head] (after qk-norm: 8 max|gamma_q| max|gamma_k|), so
  // the fixed offset bound replaces the running maximum: p = exp(s - bound), no max chain, no rescale, no branch.
  // OPT bit 4 (with bit 3, bf16 only): q arrives pre-scaled by log2(e)/8 (the QKV epilogue, q_mul = log2 e), so a score IS the exp2
  // argument, and because |score| <= bound * log2(e) <= 58 the offset is dropped altogether: p = exp2(score) lies in
  // [2^-58, 2^58], far inside the range of fp32 and bf16, and softmax is invariant to the common factor.  No per-score FMA:
  // 848 -> 917 TF per part, 952 -> 1032 per sample (r01 run 43).  Measured and rejected on top of it (run 44): row sums from the
  // matrix pipe (a third O tile against an all-ones V^T: 4 more MFMAs instead of 16 v_pk_add_f32 per key tile) -> 903 / 969.
  float mrun = (OPT & 16) ? 0.f : (OPT & 8) ? bound[head] * 8.0f : -1e30f, lsum = 0.f;
  const float c = 0.125f * 1.44269504088896340736f;   // 1/sqrt(64) * log2(e)

  const int b_first = seg0 >> 6;
  const int ntile = ((seg1 - 1) >> 6) - b_first + 1;
  // ---- register staging (DMA = false): 512 threads, one 16-byte chunk of K and one of V^T per thread per tile
  const int srow = tid >> 3, sch = (tid & 7) * 8;
  const int soff = srow * HLD + sch;
  uint4 rk = make_uint4(0, 0, 0, 0), rv = rk;
#define HATT_LOAD(T)                                                                                 \
  {                                                                                                  \
    const int blk_ = b_first + (T);                                                                  \
    int tok_ = blk_ * 64 + srow;                                                                     \
    tok_ = tok_ < TP ? tok_ : TP - 1;                                                                \
    rk = *reinterpret_cast<const uint4*>(Kg + (size_t)tok_ * 64 + sch);                              \
    rv = *reinterpret_cast<const uint4*>(Vg + ((size_t)blk_ * 64 + srow) * 64 + sch);                \
  }
#define HATT_STORE(BUF)                                                                              \
  *reinterpret_cast<uint4*>(Ks + (BUF) * (HKV * HLD) + soff) = rk;                                   \
  *reinterpret_cast<uint4*>(Vs + (BUF) * (HKV * HLD) + soff) = rv;
  // ---- LDS-DMA (DMA = true): wave w stages rows 8w .. 8w+7 of the K tile and of the V^T tile; lane -> (row 8w + lane/8, physical slot lane%8)
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

  if (DMA) {
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_)
      if (s_ < ntile) { HATT_DMA(s_, s_) }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the Q fragments have landed too -- tell the compiler (a use of every fragment), or it waits for them with vmcnt(3..0) inside
    // the key loop, where those waits would drain the DMA pieces of the NEXT tile it does not know about
#pragma unroll
    for (int s = 0; s < 4; ++s) { uint4 q_ = __builtin_bit_cast(uint4, qf[s]); asm volatile("" : "+v"(q_.x), "+v"(q_.y), "+v"(q_.z), "+v"(q_.w)); qf[s] = __builtin_bit_cast(T8, q_); }
  } else {
    HATT_LOAD(0)
    HATT_STORE(0)
  }
  __syncthreads();

  for (int t = 0; t < ntile; ++t) {
    const int cur = t & (NS - 1);
    const bool more = (t + 1) < ntile;
    const bool ahead = (t + NS - 1) < ntile;      // the tile this iteration requests: NS - 1 ahead, into the stage tile t - 1 left at the last barrier
    if (DMA) { if (ahead) { HATT_DMA(t + NS - 1, (t + NS - 1) & (NS - 1)) } } else if (more) { HATT_LOAD(t + 1) }

    if (wave_active) {
      // ---- S^T = K Q^T : two 32-key sub-tiles x 32 queries
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
      const int swz = (l31 >> 1) & 7;                 // DMA layout: slot ^ ((row >> 1) & 7), the same for rows l31 and 32 + l31
      const u16* kp = Ks + cur * (HKV * LDR) + l31 * LDR + (DMA ? 0 : 8 * hi);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int ko = DMA ? ((2 * s + hi) ^ swz) * 8 : 16 * s;
        const T8 k0 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(kp + ko));
        const T8 k1 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(kp + 32 * LDR + ko));
        s0 = H16<DT>::mfma(k0, qf[s], s0);
        s1 = H16<DT>::mfma(k1, qf[s], s1);
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
      // ---- softmax, lane-local except one cross-half max (online form only)
      if (!(OPT & 8)) {
        float mx;
        if (OPT & 1) {
          float ma = hmax3(s0[0], s0[1], s0[2]), mb = hmax3(s1[0], s1[1], s1[2]);   // two independent v_max3_f32 chains
#pragma unroll
          for (int r = 3; r < 15; r += 2) { ma = hmax3(ma, s0[r], s0[r + 1]); mb = hmax3(mb, s1[r], s1[r + 1]); }
          mx = hmax3(ma, mb, fmaxf(s0[15], s1[15]));
        } else {
          mx = fmaxf(s0[0], s1[0]);
#pragma unroll
          for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
        }
        mx = h_xhalf_max(mx);
        // deferred rescale (OPT bit 1): keep the running maximum as long as no row of the wave grew by more than DEFER_THR in the
        // exponent (P <= 2^DEFER_THR: no overflow in bf16 / fp16 / the fp32 sums); O and l are rescaled only then.
        // Textbook order: the decision precedes this tile's exponentials and follows the previous tile's P*V.
        if (!(OPT & 2) || !__all((mx - mrun) * c <= DEFER_THR)) {
          const float mnew = fmaxf(mrun, mx);
          const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * c);
          mrun = mnew;
          lsum *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
      }
      {
        // exponent FMA and the row sum on the packed fp32 pipe (v_pk_fma_f32 / v_pk_add_f32: two lanes' worth per issue slot)
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
          a.x = __builtin_amdgcn_exp2f(a.x); a.y = __builtin_amdgcn_exp2f(a.y);
          b.x = __builtin_amdgcn_exp2f(b.x); b.y = __builtin_amdgcn_exp2f(b.y);
          s0[2 * k] = a.x; s0[2 * k + 1] = a.y;
          s1[2 * k] = b.x; s1[2 * k + 1] = b.y;
          ps2 += a + b;
        }
        lsum += ps2.x + ps2.y;
      }
      // ---- O^T += V^T P^T : key step ks contracts the keys held in registers 8(ks&1)..+7 of sub-tile ks>>1
      const u16* vp = Vs + cur * (HKV * LDR) + l31 * LDR + (DMA ? 0 : 8 * hi);
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
    }

    if (more && !DMA) { HATT_STORE(cur ^ 1) }
    if (DMA) {      // this wave's two pieces of tile t + 1 have landed: loads retire in order, the NS - 2 younger tiles may stay in flight
      if (NS > 2 && ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NS - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }

  if (!wave_active) return;
  // ---- normalise and store: lane owns query l31; register r of tile e is d = 32e + crow(r, hi): groups of 4 contiguous d.
  // every wave of the block is past the last tile's barrier: the K / V^T buffers are free.  Slab of this wave: [32 queries][72]
  const float inv = 1.0f / h_xhalf_sum(lsum);
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
}

// ---- Few-token calls, second form (round 6): KEY GROUPS inside the block ------------------------------------------------------------------
// With the ring above a demo pair's launch costs what ONE wave's chain over all key tiles of its segment costs: 16 MFMAs (~0.2 us) and the
// ~250 VALU instructions of the softmax of a 32 x 64 score tile (~0.4 us) per tile, strictly one after the other because each SIMD holds one
// working wave (128-row items: 4 of the 8 waves of a block have rows) -- smaller items add blocks but leave that chain as it is (64-row
// items measured no better, r06 call 9), and a split over blocks pays for fp32 partial planes in HBM (see the notes below).  Here the 8
// waves of a block are QW = 8 / KG query waves x KG key groups: a work item is 32 QW rows (64 for KG = 4, 128 for KG = 2), the block stages KG
// key tiles per step (8 tile slots = 128 KB of LDS, NSTG = 8 / KG stages of KG tiles, requested NSTG - 1 steps ahead) and group g takes tile
// KG j + g of step j.  Every wave's chain is 1 / KG of the tiles, every SIMD holds TWO working waves (one multiplies while the other is in
// its softmax), and the KG partial (O, l, m) of a query meet in LDS after the last tile: group 0 adds them in group order (bounded softmax:
// a plain sum -- all groups use the same fixed offset; online softmax: rescaled to the largest running maximum).  No partial leaves the CU.
// Deterministic; NOT bit-identical to the unsplit kernel (fp32 summation order over the keys of a row), so it has its own tests.
template <int DT, int OPT, int KG>
__global__ __launch_bounds__(512, 2) void attention_h16_kgroup_kernel(const u16* __restrict__ qk, const u16* __restrict__ vt, int vt_nblk,
                                                                      u16* __restrict__ out, int TP, int heads,
                                                                      const AttnWorkItem* __restrict__ items, const float* __restrict__ bound) {
  typedef typename H16<DT>::T8 T8;
  static_assert(KG == 2 || KG == 4, "two or four key groups");
  constexpr int QW = 8 / KG;                 // query waves of the block (work item = 32 QW rows)
  constexpr int NSTG = 8 / KG;               // stages of KG tiles: 8 tile slots of K and 8 of V^T = 128 KB
  constexpr int TILE = HKV * 64;             // 16-bit elements of one K (or V^T) tile, 128-byte rows, slots XOR-swizzled
  constexpr int MERGE_OFF = 8 * 32 * HLD * 2;   // bytes: the merge records start behind the 8 output slabs
  static_assert(MERGE_OFF + (KG - 1) * QW * 34 * 64 * 4 <= 16 * TILE * 2, "merge records must fit");
  extern __shared__ __attribute__((aligned(16))) u16 smem_kg[];
  u16* Ks = smem_kg;                 // [8 slots][64 keys][64]
  u16* Vs = smem_kg + 8 * TILE;      // [8 slots][64 d][64]   (columns = vt_pos of the key)

  const int tid = threadIdx.x;
  const int head = blockIdx.x % heads;
  const AttnWorkItem it = items[blockIdx.x / heads];
  const int len = it.seg_len;
  if (len <= 0) return;
  const int lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int seg0 = it.seg_start, seg1 = it.seg_start + len;
  const int qwv = wave % QW, grp = wave / QW;

  const u16* Qg = qk + (size_t)head * TP * 64;
  const u16* Kg = qk + (size_t)(heads + head) * TP * 64;
  const u16* Vg = vt + (size_t)head * vt_nblk * (64 * 64);

  const int qw0 = it.q0 + qwv * 32;
  const bool has_rows = qw0 < len;

  T8 qf[4];
  {
    int q = qw0 + l31;
    q = q < len ? q : len - 1;
    const u16* qp = Qg + (size_t)(seg0 + q) * 64 + 8 * hi;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(qp + 16 * s));
  }
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float mrun = (OPT & 16) ? 0.f : (OPT & 8) ? bound[head] * 8.0f : -1e30f, lsum = 0.f;
  const float c = 0.125f * 1.44269504088896340736f;

  const int b_first = seg0 >> 6;
  const int ntile = ((seg1 - 1) >> 6) - b_first + 1;
  const int nstep = (ntile + KG - 1) / KG;
  // staging as in the kernel above: wave w owns rows 8w .. 8w+7 of every K tile and every V^T tile of a step (2 KG pieces of 1 KB per step).
  // A step beyond the segment's last tile re-requests that tile: every step issues the same number of pieces, so the counted wait holds.
  const int drow = wave * 8 + (lane >> 3);
  const int dls = ((tid & 7) ^ ((drow >> 1) & 7)) * 8;
  const unsigned lds_k = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) u16*)Ks + (unsigned)wave * 1024u);
  const unsigned lds_v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) u16*)Vs + (unsigned)wave * 1024u);
#define HKG_DMA_STEP(J)                                                                               \
  {                                                                                                   \
    _Pragma("unroll") for (int gg_ = 0; gg_ < KG; ++gg_) {                                            \
      int t_ = (J) * KG + gg_;                                                                        \
      t_ = t_ < ntile ? t_ : ntile - 1;                                                               \
      const int blk_ = b_first + t_;                                                                  \
      int tok_ = blk_ * 64 + drow;                                                                    \
      tok_ = tok_ < TP ? tok_ : TP - 1;                                                               \
      const unsigned slot_ = (unsigned)((((J) % NSTG) * KG + gg_) * (TILE * 2));                      \
      HATT_DMA1(Kg + (size_t)tok_ * 64 + dls, lds_k + slot_)                                          \
      HATT_DMA1(Vg + ((size_t)blk_ * 64 + drow) * 64 + dls, lds_v + slot_)                            \
    }                                                                                                 \
  }
#pragma unroll
  for (int s_ = 0; s_ < NSTG - 1; ++s_)
    if (s_ < nstep) HKG_DMA_STEP(s_)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int s = 0; s < 4; ++s) { uint4 q_ = __builtin_bit_cast(uint4, qf[s]); asm volatile("" : "+v"(q_.x), "+v"(q_.y), "+v"(q_.z), "+v"(q_.w)); qf[s] = __builtin_bit_cast(T8, q_); }
  __syncthreads();

  const int swz = (l31 >> 1) & 7;
  for (int j = 0; j < nstep; ++j) {
    const bool ahead = (j + NSTG - 1) < nstep;
    if (ahead) HKG_DMA_STEP(j + NSTG - 1)
    const int t = j * KG + grp;
    if (has_rows && t < ntile) {
      const int cur = (j % NSTG) * KG + grp;
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
      const u16* kp = Ks + cur * TILE + l31 * 64;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int ko = ((2 * s + hi) ^ swz) * 8;
        const T8 k0 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(kp + ko));
        const T8 k1 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(kp + 32 * 64 + ko));
        s0 = H16<DT>::mfma(k0, qf[s], s0);
        s1 = H16<DT>::mfma(k1, qf[s], s1);
      }
      const int tile0 = (b_first + t) * 64;
      if (tile0 < seg0 || tile0 + 64 > seg1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kg = tile0 + mfma32_crow(r, hi);
          s0[r] = (kg >= seg0 && kg < seg1) ? s0[r] : -1e30f;
          s1[r] = (kg + 32 >= seg0 && kg + 32 < seg1) ? s1[r] : -1e30f;
        }
      }
      if (!(OPT & 8)) {        // online softmax: v_max3 row maxima, deferred rescale (as above)
        float ma = hmax3(s0[0], s0[1], s0[2]), mb = hmax3(s1[0], s1[1], s1[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) { ma = hmax3(ma, s0[r], s0[r + 1]); mb = hmax3(mb, s1[r], s1[r + 1]); }
        const float mx = h_xhalf_max(hmax3(ma, mb, fmaxf(s0[15], s1[15])));
        if (!__all((mx - mrun) * c <= DEFER_THR)) {
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
          if (!(OPT & 16)) {
            a = __builtin_elementwise_fma(a, c2, nmc2);
            b = __builtin_elementwise_fma(b, c2, nmc2);
          }
          a.x = __builtin_amdgcn_exp2f(a.x); a.y = __builtin_amdgcn_exp2f(a.y);
          b.x = __builtin_amdgcn_exp2f(b.x); b.y = __builtin_amdgcn_exp2f(b.y);
          s0[2 * k] = a.x; s0[2 * k + 1] = a.y;
          s1[2 * k] = b.x; s1[2 * k + 1] = b.y;
          ps2 += a + b;
        }
        lsum += ps2.x + ps2.y;
      }
      const u16* vp = Vs + cur * TILE + l31 * 64;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int rb = 8 * (ks & 1);
        T8 pb;
        if ((ks >> 1) == 0)
          pb = h16_pack8<DT>(s0[rb + 0], s0[rb + 1], s0[rb + 2], s0[rb + 3], s0[rb + 4], s0[rb + 5], s0[rb + 6], s0[rb + 7]);
        else
          pb = h16_pack8<DT>(s1[rb + 0], s1[rb + 1], s1[rb + 2], s1[rb + 3], s1[rb + 4], s1[rb + 5], s1[rb + 6], s1[rb + 7]);
        const int vo = ((2 * ks + hi) ^ swz) * 8;
        const T8 v0 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(vp + vo));
        const T8 v1 = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(vp + 32 * 64 + vo));
        o0 = H16<DT>::mfma(v0, pb, o0);
        o1 = H16<DT>::mfma(v1, pb, o1);
      }
    }
    if (NSTG > 2 && ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * KG * (NSTG - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#undef HKG_DMA_STEP

  // ---- the KG partials of a query meet in LDS (the K / V^T slots are free: every wave is past the last barrier).  Record of a wave:
  // 34 values per lane (o0, o1, l, m), value-major, so that a wave's 64 lanes write / read 256 contiguous bytes.
  float* rec = reinterpret_cast<float*>(reinterpret_cast<char*>(smem_kg) + MERGE_OFF);
  if (grp > 0 && has_rows) {
    float* wr = rec + (size_t)((grp - 1) * QW + qwv) * (34 * 64) + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) { wr[r * 64] = o0[r]; wr[(16 + r) * 64] = o1[r]; }
    wr[32 * 64] = lsum;
    wr[33 * 64] = mrun;
  }
  __syncthreads();
  if (grp > 0 || !has_rows) return;
  if (OPT & 8) {               // bounded softmax: one fixed offset for every group -- plain sums, in group order
#pragma unroll
    for (int g = 1; g < KG; ++g) {
      const float* rd = rec + (size_t)((g - 1) * QW + qwv) * (34 * 64) + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] += rd[r * 64]; o1[r] += rd[(16 + r) * 64]; }
      lsum += rd[32 * 64];
    }
  } else {                     // online softmax: everything is brought to the largest running maximum of the KG groups
    float mg[KG], mall = mrun;
    mg[0] = mrun;
#pragma unroll
    for (int g = 1; g < KG; ++g) { mg[g] = rec[(size_t)((g - 1) * QW + qwv) * (34 * 64) + 33 * 64 + lane]; mall = fmaxf(mall, mg[g]); }
    const float a0 = __builtin_amdgcn_exp2f((mg[0] - mall) * c);
    lsum *= a0;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= a0; o1[r] *= a0; }
#pragma unroll
    for (int g = 1; g < KG; ++g) {
      const float* rd = rec + (size_t)((g - 1) * QW + qwv) * (34 * 64) + lane;
      const float ag = __builtin_amdgcn_exp2f((mg[g] - mall) * c);
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] = fmaf(rd[r * 64], ag, o0[r]); o1[r] = fmaf(rd[(16 + r) * 64], ag, o1[r]); }
      lsum = fmaf(rd[32 * 64], ag, lsum);
    }
  }
  const float inv = 1.0f / h_xhalf_sum(lsum);
  u16* slab = smem_kg + wave * (32 * HLD);
  u16* wp = slab + l31 * HLD + 4 * hi;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    *reinterpret_cast<uint2*>(wp + 8 * g) =
        h16_pack4<DT>(o0[4 * g + 0] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
    *reinterpret_cast<uint2*>(wp + 32 + 8 * g) =
        h16_pack4<DT>(o1[4 * g + 0] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (lane >> 3) + 8 * i, piece = lane & 7;
    const uint4 v = *reinterpret_cast<const uint4*>(slab + row * HLD + piece * 8);
    if (qw0 + row < len)
      *reinterpret_cast<uint4*>(out + (size_t)(seg0 + qw0 + row) * (heads * 64) + head * 64 + piece * 8) = v;
  }
}

// Kernel choice per LAUNCH (round 3): per-head logit bounds supplied (the caller guarantees q.k/8 <= bound[h] <= 40) -> the bounded,
// offset-free softmax (bf16: on pre-scaled q when the model path asks for it); no bounds, or fp16 (whose exponent range cannot
// hold 2^58) -> the online softmax with v_max3 row maxima and deferred rescale.  Both are the same kernel template; the output
// leaves as whole rows through an LDS slab.  The other schedules of rounds 1-2 (ping-pong, two software-pipelined kernels,
// persistent blocks, rotated key walk, 512-query blocks -- all measured equal or slower) are in the history at 72efb73.
// RAP_ABLATION_BUILD only: rap_set_tuning(3, 5) forces the online softmax even with bounds (A/B of the bounded kernel).
// Few-token calls (one pair of 2 x 1024 points: 64 blocks for 256 CUs) -- two ways of filling the chip were measured in round 3 and NOT
// kept: (a) 4 / 2 blocks of 64 / 128 queries per work item (bit-identical results): 18.4 vs 17.6 ms per bf16 call, 53.5 vs 50.8 at
// 2 x 2048 (call 34) -- a wave then issues 4 / 2 times the LDS-DMA pieces per tile, and their issue cost is the wave's chain; (b) the fp32
// kernel's split over key ranges with fp32 partial planes and a combine pass: 17.5 vs 17.7 ms and 54.6 vs 51.4 (call 35) -- the kernel
// drops from 23.8 to 19.2 us per launch, the combine pass adds 5.3.  At that size every kernel of the layer sits at the 5-13 us launch floor.
// Round 6 (GPU call 6, source at commit da4c269,
// profiles/r06_c6_h16_split_kv_in_kernel_merge_*): (c) the same split with the merge INSIDE the kernel -- every block releases its fp32
// partial (agent scope), the last arrival per work item re-reads all of them in split order and stores the result, no combine launch: correct
// and bit-stable (134 tests), but 43.4 us per launch against 24.1 (21.0 vs 16.5 ms per bf16 call).  Two reasons, both in the CDNA4 guide:
// a 4-way split leaves 4 x 64 KB of fp32 partials per work item for ONE block to read back serially (~1 us per 16 KB), and the arrival
// flag as a second __shared__ object makes hipcc drain the LDS-DMA stream (vmcnt(0)) in front of every fragment read of the key loop.
// The partials of a split cost 8 x the bytes of the 16-bit output they replace: at this size the attention launch stays unsplit.
rap_tuning_t g_rap_attn_h16_variant = 0;
rap_tuning_t g_rap_attn_h16_dma = 1;      // tuning key 13: K / V^T tiles by LDS-DMA (1, default) or staged through registers (0)

// Query rows per work item and key groups per block: 256 rows / one group, or -- few-token calls (tuning key 20) -- smaller items where 256-row
// items leave most of the CUs without a block.  rows = align_up(TP, 256) of the call (0: the kernel-level entry points).
//   key 20 = 1 (default): the rule below;  0: 256 rows, two stages (round 5);  2: 128-row items + four-stage ring up to 4 096 rows, no key
//   groups (the first form of round 6: bit-identical to round 5);  64 / 128: that item size + ring for every call of at most 8 192 rows (A/B);
//   66 / 130: 64-row items x 4 key groups / 128-row items x 2 key groups for every call of at most 8 192 rows (A/B)
rap_tuning_t g_rap_attn_h16_small = 1;
static void attention_h16_plan(long rows, int* bq, int* kg) {
  const int mode = g_rap_attn_h16_small;
  *bq = 256; *kg = 1;
  if (!mode || !g_rap_attn_h16_dma || rows <= 0 || rows > 8192) return;
  if (mode == 64 || mode == 128) { *bq = mode; return; }
  if (mode == 66) { *bq = 64; *kg = 4; return; }
  if (mode == 130) { *bq = 128; *kg = 2; return; }
  // r06 call 9 (bf16, ms per call; 64 / 128 / 256-row items, no key groups): 2 048 rows 16.4 / 16.2 / 17.6, 4 096 rows 48.9 / 43.3 / 47.0,
  // 8 000 rows 110 / 80.8 / 77.5 -- 128-row items up to 4 096 rows, 256 above
  if (rows > 4096) return;
  *bq = 128;
  if (mode == 2) return;
  if (rows <= 2048) { *bq = 64; *kg = 4; } else { *kg = 2; }
}
int attention_h16_block_queries(int, long rows) { int bq, kg; attention_h16_plan(rows, &bq, &kg); return bq; }
int attention_h16_key_groups(int, long rows) { int bq, kg; attention_h16_plan(rows, &bq, &kg); return kg; }
bool attention_h16_forced() { const int m = g_rap_attn_h16_small; return m == 64 || m == 128 || m == 66 || m == 130; }

// the model path asks before it runs qk-norm: pre-scaled q only feeds the bounded bf16 kernel
bool attention_h16_wants_prescaled_q(int dtype, bool bounded) {
#ifdef RAP_ABLATION_BUILD
  if (g_rap_attn_h16_variant == 5) return false;
#endif
  return dtype == RAP_DT_BF16 && bounded;
}

#define HKG_LDS_BYTES (16 * HKV * 64 * 2)      // 8 K + 8 V^T tile slots
template <int DT, int OPT, int KG>
static int launch_kgroup(hipStream_t stream, const u16* qk, const u16* vt, int vt_nblk, u16* out, int TP, int heads, const AttnWorkItem* items,
                         int max_items, const float* bound) {
  // more than the 64 KB a kernel may use without asking; set per launch (as attn_x2.hip does): the attribute belongs to the current device
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_h16_kgroup_kernel<DT, OPT, KG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          HKG_LDS_BYTES) != hipSuccess) {
    rap_set_last_hip_error((int)hipGetLastError());
    return RAP_ERR_HIP;
  }
  hipLaunchKernelGGL((attention_h16_kgroup_kernel<DT, OPT, KG>), dim3(max_items * heads), dim3(512), HKG_LDS_BYTES, stream, qk, vt, vt_nblk, out, TP,
                     heads, items, bound);
  return RAP_OK;
}

int launch_attention_h16(hipStream_t stream, int dtype, const u16* qk, const u16* vt, int vt_nblk, u16* out, int TP,
                         int heads, const AttnWorkItem* items, int max_items, const float* bound, int q_prescaled, int bq, int kg) {
  if (max_items <= 0 || TP <= 0) return RAP_OK;
  if (heads <= 0 || vt_nblk * 64 < TP || (bq != 64 && bq != 128 && bq != 256)) return RAP_ERR_INVALID;
  if (kg != 1 && !(kg == 4 && bq == 64) && !(kg == 2 && bq == 128)) return RAP_ERR_INVALID;      // 8 waves = bq / 32 query waves x kg key groups
  // few-token work lists (small items: one block per CU, nothing hides the next tile's round trip) take the four-stage ring.  For 256-row
  // items with at most two blocks per CU (8 000 rows) the ring measured 77.5 vs 76.9 ms per call (r06 call 10): two stages there.
  const bool ring = g_rap_attn_h16_dma && bq < 256;
  if (q_prescaled && !(bound && attention_h16_wants_prescaled_q(dtype, true))) return RAP_ERR_INVALID;
#ifdef RAP_ABLATION_BUILD
  if (g_rap_attn_h16_variant == 5) bound = nullptr;
#endif
#define HATT_LAUNCH(DTV, OPTV)                                                                                                            \
  {                                                                                                                                       \
    if (kg == 4) {                                                                                                                        \
      if (int rc_ = launch_kgroup<DTV, OPTV, 4>(stream, qk, vt, vt_nblk, out, TP, heads, items, max_items, bound)) return rc_;            \
    } else if (kg == 2) {                                                                                                                 \
      if (int rc_ = launch_kgroup<DTV, OPTV, 2>(stream, qk, vt, vt_nblk, out, TP, heads, items, max_items, bound)) return rc_;            \
    }                                                                                                                                     \
    else if (ring)                                                                                                                        \
      hipLaunchKernelGGL((attention_h16_kernel<DTV, OPTV, true, 4>), dim3(max_items * heads), dim3(512), 0, stream, qk, vt, vt_nblk, out, TP, heads, items, bound, bq); \
    else if (g_rap_attn_h16_dma)                                                                                                          \
      hipLaunchKernelGGL((attention_h16_kernel<DTV, OPTV, true>), dim3(max_items * heads), dim3(512), 0, stream, qk, vt, vt_nblk, out, TP, heads, items, bound, bq); \
    else                                                                                                                                  \
      hipLaunchKernelGGL((attention_h16_kernel<DTV, OPTV, false>), dim3(max_items * heads), dim3(512), 0, stream, qk, vt, vt_nblk, out, TP, heads, items, bound, bq); \
  }
  if (dtype == RAP_DT_BF16) {
    if (bound && q_prescaled) HATT_LAUNCH(RAP_DT_BF16, 24)
    else if (bound) HATT_LAUNCH(RAP_DT_BF16, 8)
    else HATT_LAUNCH(RAP_DT_BF16, 3)
  } else if (dtype == RAP_DT_F16) {
    HATT_LAUNCH(RAP_DT_F16, 3)
  } else {
    return RAP_ERR_INVALID;
  }
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
