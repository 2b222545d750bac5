// fp32 GEMM on the CDNA4 matrix cores:  C (M,N) = A (M,K) * W (N,K)^T  + fused epilogue.
//
// Replaces the nn.Linear calls on the reference's velocity-network path
// (flow_model/layer.py:73-74,81-82,89 ; embedding.py:179 ; point_cloud_dit.py:111-117).
//
// Design (gfx950 only):
//  * v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 64 cyc/SIMD, one VGPR per operand.
//  * 128x128 block tile, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles (64 accumulator regs).
//  * BK = 32: every A/W row contributes one full 128-byte line per k-tile.
//  * LDS image [row][36 floats] (144-B row stride = 9 sixteen-byte slots: the 16 rows of each
//    ds_read_b128 lane group land on 16 distinct slots -> conflict free).
//  * The contraction order inside a k-tile is permuted so that one ds_read_b128 feeds four MFMAs:
//    for k-group g (8 columns) lanes 0-31 read columns 8g..8g+3 and lanes 32-63 columns 8g+4..8g+7;
//    MFMA step s then contracts the column pair {8g+s, 8g+4+s}.  A sum is order independent up to
//    fp32 rounding, and the same permutation is applied to A and W.
//  * global -> register -> LDS staging, double-buffered LDS, one barrier per k-tile; the loads of
//    tile t+1 are issued before the 64 MFMAs of tile t.
//  * 1-D grid, XCD-aware remap, n-tile fastest: all n-tiles of one 128-row A panel run back to
//    back on one XCD (A panel stays in that XCD's L2; W streams from L2 / Infinity Cache).
#include <type_traits>
#include "half.h"
#include "kernels.h"

#define GBM 128
#define GBN 128
#define GBK 32
#define GLD 36  // LDS row stride in floats

// Fused MultiHeadRMSNorm for one 64-column wave tile (= one head of q or k) of the QKV projection: acc[mi][ni][r] is row
// m = mw + 32 mi + crow(r, hi), column 32 ni + l31.  The row norm is a sum over the 32 lanes of a half-wave and the two column
// tiles (5 xor-shuffles per row); the scaling repeats qknorm_kernel's operation order (x / nrm * gamma * 8), so the result differs
// from GEMM + qknorm only through the summation order of the 64 squares.
template <int TMI>
__device__ __forceinline__ void qkv_store_normalised(const GemmParams& p, f32x16 (&acc)[TMI][2], int mw, int nw, int hi, int l31) {
  const int dmodel = p.heads * 64;
  const int c = nw / dmodel;
  const int h = (nw - c * dmodel) >> 6;
  const float* gam = (c == 0 ? p.gamma_q : p.gamma_k) + h * 64;
  const float g0 = gam[l31], g1 = gam[32 + l31];
  float* plane = p.C + ((size_t)c * p.heads + h) * p.M * 64;
#pragma unroll
  for (int mi = 0; mi < TMI; ++mi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s = acc[mi][0][r] * acc[mi][0][r] + acc[mi][1][r] * acc[mi][1][r];
      // sum over the 32 lanes of the half-wave (= the 64 columns of the head): four DPP steps inside the VALU (quad xor 1, quad xor 2,
      // mirror within 8, mirror within 16 -- mirrored partners hold the same partial sums, so every lane ends with the total) and one
      // ds_swizzle across the two 16-lane rows, instead of five ds_bpermute round trips through the LDS queue (r02: 320 per wave tile)
      s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, true));
      s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, true));
      s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x141, 0xF, 0xF, true));
      s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x140, 0xF, 0xF, true));
      s += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, s), 0x401F));
      const float nrm = fmaxf(sqrtf(s), 1e-12f);
      const int m = mw + mi * 32 + mfma32_crow(r, hi);
      if (m < p.M) {
        plane[(size_t)m * 64 + l31] = acc[mi][0][r] / nrm * g0 * 8.0f;
        plane[(size_t)m * 64 + 32 + l31] = acc[mi][1][r] / nrm * g1 * 8.0f;
      }
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * GBM * GLD];
  float* As = smem;                      // [2][128][36]
  float* Bs = smem + 2 * GBM * GLD;      // [2][128][36]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int hi = lane >> 5;
  const int l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  const int nt = p.N / GBN;
  const int mt = (p.M + GBM - 1) / GBM;
  const int logical = xcd_remap(blockIdx.x, mt * nt);
  const int m0 = (logical / nt) * GBM;
  const int n0 = (logical % nt) * GBN;

  // per-thread staging coordinates: 4 float4 of A and 4 of W per k-tile
  const int srow = tid >> 3;         // 0..31 (+32*i)
  const int sc4 = (tid & 7) * 4;     // float column inside the k-tile
  const float* a_ptr[4];
  const float* w_ptr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = m0 + srow + 32 * i;
    r = r < p.M ? r : p.M - 1;
    a_ptr[i] = p.A + (size_t)r * p.lda + sc4;
    w_ptr[i] = p.W + (size_t)(n0 + srow + 32 * i) * p.ldw + sc4;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;   // named registers: arrays under a branch end up in scratch
  const int nk = p.K / GBK;
  const int st_off = srow * GLD + sc4;

#define GEMM_LOAD_TILE(KT)                                                            \
  ra0 = *reinterpret_cast<const float4*>(a_ptr[0] + (size_t)(KT) * GBK);              \
  ra1 = *reinterpret_cast<const float4*>(a_ptr[1] + (size_t)(KT) * GBK);              \
  ra2 = *reinterpret_cast<const float4*>(a_ptr[2] + (size_t)(KT) * GBK);              \
  ra3 = *reinterpret_cast<const float4*>(a_ptr[3] + (size_t)(KT) * GBK);              \
  rw0 = *reinterpret_cast<const float4*>(w_ptr[0] + (size_t)(KT) * GBK);              \
  rw1 = *reinterpret_cast<const float4*>(w_ptr[1] + (size_t)(KT) * GBK);              \
  rw2 = *reinterpret_cast<const float4*>(w_ptr[2] + (size_t)(KT) * GBK);              \
  rw3 = *reinterpret_cast<const float4*>(w_ptr[3] + (size_t)(KT) * GBK);
#define GEMM_STORE_TILE(BUF)                                                          \
  *reinterpret_cast<float4*>(&As[(BUF) * (GBM * GLD) + st_off + 0 * 32 * GLD]) = ra0; \
  *reinterpret_cast<float4*>(&As[(BUF) * (GBM * GLD) + st_off + 1 * 32 * GLD]) = ra1; \
  *reinterpret_cast<float4*>(&As[(BUF) * (GBM * GLD) + st_off + 2 * 32 * GLD]) = ra2; \
  *reinterpret_cast<float4*>(&As[(BUF) * (GBM * GLD) + st_off + 3 * 32 * GLD]) = ra3; \
  *reinterpret_cast<float4*>(&Bs[(BUF) * (GBN * GLD) + st_off + 0 * 32 * GLD]) = rw0; \
  *reinterpret_cast<float4*>(&Bs[(BUF) * (GBN * GLD) + st_off + 1 * 32 * GLD]) = rw1; \
  *reinterpret_cast<float4*>(&Bs[(BUF) * (GBN * GLD) + st_off + 2 * 32 * GLD]) = rw2; \
  *reinterpret_cast<float4*>(&Bs[(BUF) * (GBN * GLD) + st_off + 3 * 32 * GLD]) = rw3;
#define GEMM_COMPUTE_TILE(BUF)                                                                        \
  {                                                                                                   \
    const float* Ac = As + (BUF) * (GBM * GLD);                                                       \
    const float* Bc = Bs + (BUF) * (GBN * GLD);                                                       \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                   \
      const float4 a0 = *reinterpret_cast<const float4*>(&Ac[a_off + 8 * g]);                         \
      const float4 a1 = *reinterpret_cast<const float4*>(&Ac[a_off + 32 * GLD + 8 * g]);              \
      const float4 b0 = *reinterpret_cast<const float4*>(&Bc[b_off + 8 * g]);                         \
      const float4 b1 = *reinterpret_cast<const float4*>(&Bc[b_off + 32 * GLD + 8 * g]);              \
      GEMM_MFMA4(a0.x, a1.x, b0.x, b1.x)                                                              \
      GEMM_MFMA4(a0.y, a1.y, b0.y, b1.y)                                                              \
      GEMM_MFMA4(a0.z, a1.z, b0.z, b1.z)                                                              \
      GEMM_MFMA4(a0.w, a1.w, b0.w, b1.w)                                                              \
    }                                                                                                 \
  }
#define GEMM_MFMA4(A0, A1, B0, B1)                                                   \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B0, acc[0][0], 0, 0, 0);      \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B1, acc[0][1], 0, 0, 0);      \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B0, acc[1][0], 0, 0, 0);      \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B1, acc[1][1], 0, 0, 0);

  const int a_off = (wm * 64 + l31) * GLD + 4 * hi;
  const int b_off = (wn * 64 + l31) * GLD + 4 * hi;

  GEMM_LOAD_TILE(0)
  GEMM_STORE_TILE(0)
  __syncthreads();

  // main loop: prefetch tile kt+1 into registers, 64 MFMAs on tile kt, park the prefetch in the other buffer
  int kt = 0;
  for (; kt + 1 < nk; ++kt) {
    const int cur = kt & 1;
    GEMM_LOAD_TILE(kt + 1)
    __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ABOVE the MFMAs (hipcc otherwise sinks it to its use)
    GEMM_COMPUTE_TILE(cur)
    __builtin_amdgcn_sched_barrier(0);
    GEMM_STORE_TILE(cur ^ 1)
    __syncthreads();
  }
  GEMM_COMPUTE_TILE(kt & 1)

  // ---------------- epilogue ----------------
  const int mw = m0 + wm * 64;
  const int nw = n0 + wn * 64;
  if (EPI == EPI_GEGLU) {
    // acc[mi][0] = value columns, acc[mi][1] = gate columns of the same 32 outputs
    const int nout = (nw >> 1) + l31;
    const float bh = p.bias ? p.bias[nw + l31] : 0.f;
    const float bg = p.bias ? p.bias[nw + 32 + l31] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mw + mi * 32 + mfma32_crow(r, hi);
        if (m < p.M) {
          const float h = acc[mi][0][r] + bh;
          const float g = acc[mi][1][r] + bg;
          const float ge = 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f));
          p.C[(size_t)m * p.ldc + nout] = h * ge;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = nw + ni * 32 + l31;
      const float bn = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mw + mi * 32 + mfma32_crow(r, hi);
        if (m >= p.M) continue;
        float v = acc[mi][ni][r] + bn;
        if (EPI == EPI_BIAS) {
          p.C[(size_t)m * p.ldc + n] = v;
        } else if (EPI == EPI_BIAS_RESID) {
          p.C[(size_t)m * p.ldc + n] = p.resid[(size_t)m * p.ldr + n] + v;
        } else if (EPI == EPI_BIAS_SILU) {
          p.C[(size_t)m * p.ldc + n] = v / (1.0f + expf(-v));
        } else if (EPI == EPI_BIAS_RELU) {
          p.C[(size_t)m * p.ldc + n] = fmaxf(v, 0.f);
        } else if (EPI == EPI_BIAS_ANCHOR) {
          const int sel = p.anchor[m] ? 1 : 0;
          p.C[(size_t)m * p.ldc + n] = v + p.anchor_emb[(size_t)sel * p.N + n];
        } else if (EPI == EPI_QKV_HEADMAJOR) {
          const int dmodel = p.heads * 64;
          const int c = n / dmodel;
          const int rem = n - c * dmodel;
          const int h = rem >> 6, j = rem & 63;
          p.C[(((size_t)c * p.heads + h) * p.M + m) * 64 + j] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Software-pipelined variant (the default).  Same tiling idea, but
//  * the MFMA operands of k-group g+1 are read from LDS into a second register set while the 16*WNT/2
//    MFMAs of group g issue (fragment double-buffering), and the first group of tile t+1 is read right
//    after the barrier while the LAST group of tile t still runs -> no exposed ds_read latency after a barrier;
//  * the prefetched tile t+1 is parked in the spare LDS buffer in the middle of tile t's MFMAs
//    (it has had 32*WNT/2 MFMAs = 2-4k cycles to arrive), not in a serial block before the barrier;
//  * WNT = 4: 128x256 block tile, each wave 64x128 (8 accumulator tiles = 128 registers), one block per CU
//    (110 KB LDS): 1.33x fewer L2->LDS bytes per flop than 128x128 and 128 MFMAs between barriers.
//    WNT = 2 keeps the 128x128 tile at two blocks per CU.
// ---------------------------------------------------------------------------------------------
//  * WMW = 4: 8 waves (4 along M x 2 along N), 256x128 block tile, one block per CU; the two waves that share a
//    SIMD (w, w+4) get different static priorities so they do not run in lock-step (see attn_f32.hip).
template <int EPI, int WNT, int WMW>
__global__ __launch_bounds__(128 * WMW, (WMW == 4 ? 2 : (WNT == 2 ? 2 : 1))) void gemm_f32_pipe_kernel(GemmParams p) {
  constexpr int BM = 64 * WMW;        // block tile along M
  constexpr int BN = 64 * WNT;        // block tile along N (two waves)
  constexpr int RP = 16 * WMW;        // rows staged per pass (8 threads per 32-float row)
  constexpr int NW4 = BN / RP;        // float4 of W staged per thread per k-tile (A: always 4)
  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * GLD];
  float* As = smem;                      // [2][BM][36]
  float* Bs = smem + 2 * BM * GLD;       // [2][BN][36]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int hi = lane >> 5;
  const int l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  if (WMW == 4) {
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
  }

  const int nt = p.N / BN;
  const int mt = (p.M + BM - 1) / BM;
  const int logical = xcd_remap(blockIdx.x, mt * nt);
  const int m0 = (logical / nt) * BM;
  const int n0 = (logical % nt) * BN;

  const int srow = tid >> 3;
  const int sc4 = (tid & 7) * 4;
  const float* a_ptr[4];
  const float* w_ptr[NW4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = m0 + srow + RP * i;
    r = r < p.M ? r : p.M - 1;
    a_ptr[i] = p.A + (size_t)r * p.lda + sc4;
  }
#pragma unroll
  for (int i = 0; i < NW4; ++i) w_ptr[i] = p.W + (size_t)(n0 + srow + RP * i) * p.ldw + sc4;

  f32x16 acc[2][WNT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Staging and fragment registers are named struct members (never arrays: hipcc promotes private arrays
  // that survive until late unrolling to LDS or scratch).
  float4 sg_a0, sg_a1, sg_a2, sg_a3, sg_w0, sg_w1, sg_w2, sg_w3, sg_w4, sg_w5, sg_w6, sg_w7;
  struct Frag { float4 a0, a1, b0, b1, b2, b3; } f0, f1;
  const int nk = p.K / GBK;
  const int st_off = srow * GLD + sc4;
  const int a_off = (wm * 64 + l31) * GLD + 4 * hi;
  const int b_off = (wn * (32 * WNT) + l31) * GLD + 4 * hi;

#define PG_LD(PTR, KT) (*reinterpret_cast<const float4*>((PTR) + (size_t)(KT) * GBK))
#define PG_LOAD(KT)                                                                    \
  sg_a0 = PG_LD(a_ptr[0], KT); sg_a1 = PG_LD(a_ptr[1], KT);                            \
  sg_a2 = PG_LD(a_ptr[2], KT); sg_a3 = PG_LD(a_ptr[3], KT);                            \
  sg_w0 = PG_LD(w_ptr[0], KT); sg_w1 = PG_LD(w_ptr[1], KT);                            \
  if constexpr (NW4 >= 4) { sg_w2 = PG_LD(w_ptr[2], KT); sg_w3 = PG_LD(w_ptr[3], KT); } \
  if constexpr (NW4 == 8) {                                                            \
    sg_w4 = PG_LD(w_ptr[4], KT); sg_w5 = PG_LD(w_ptr[5], KT);                          \
    sg_w6 = PG_LD(w_ptr[6], KT); sg_w7 = PG_LD(w_ptr[7], KT);                          \
  }
#define PG_ST(BASE, I, V) *reinterpret_cast<float4*>(&(BASE)[st_off + (I) * RP * GLD]) = (V)
#define PG_STORE(BUF)                                                                  \
  {                                                                                    \
    float* A_ = As + (BUF) * (BM * GLD);                                               \
    float* B_ = Bs + (BUF) * (BN * GLD);                                               \
    PG_ST(A_, 0, sg_a0); PG_ST(A_, 1, sg_a1); PG_ST(A_, 2, sg_a2); PG_ST(A_, 3, sg_a3); \
    PG_ST(B_, 0, sg_w0); PG_ST(B_, 1, sg_w1);                                          \
    if constexpr (NW4 >= 4) { PG_ST(B_, 2, sg_w2); PG_ST(B_, 3, sg_w3); }              \
    if constexpr (NW4 == 8) {                                                          \
      PG_ST(B_, 4, sg_w4); PG_ST(B_, 5, sg_w5); PG_ST(B_, 6, sg_w6); PG_ST(B_, 7, sg_w7); \
    }                                                                                  \
  }
#define PG_RD(BASE, OFF, I, G) (*reinterpret_cast<const float4*>(&(BASE)[(OFF) + (I) * 32 * GLD + 8 * (G)]))
#define PG_READ(BUF, G, F)                                                             \
  {                                                                                    \
    const float* A_ = As + (BUF) * (BM * GLD);                                         \
    const float* B_ = Bs + (BUF) * (BN * GLD);                                         \
    F.a0 = PG_RD(A_, a_off, 0, G); F.a1 = PG_RD(A_, a_off, 1, G);                      \
    F.b0 = PG_RD(B_, b_off, 0, G); F.b1 = PG_RD(B_, b_off, 1, G);                      \
    if constexpr (WNT == 4) { F.b2 = PG_RD(B_, b_off, 2, G); F.b3 = PG_RD(B_, b_off, 3, G); } \
  }
#define PG_MM(I, J, AV, BV) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV, BV, acc[I][J], 0, 0, 0);
#define PG_MFMA_STEP(F, C)                                                             \
  PG_MM(0, 0, F.a0.C, F.b0.C) PG_MM(0, 1, F.a0.C, F.b1.C)                              \
  if constexpr (WNT == 4) { PG_MM(0, 2, F.a0.C, F.b2.C) PG_MM(0, 3, F.a0.C, F.b3.C) }  \
  PG_MM(1, 0, F.a1.C, F.b0.C) PG_MM(1, 1, F.a1.C, F.b1.C)                              \
  if constexpr (WNT == 4) { PG_MM(1, 2, F.a1.C, F.b2.C) PG_MM(1, 3, F.a1.C, F.b3.C) }
#define PG_MFMA(F) PG_MFMA_STEP(F, x) PG_MFMA_STEP(F, y) PG_MFMA_STEP(F, z) PG_MFMA_STEP(F, w)

  PG_LOAD(0)
  PG_STORE(0)
  __syncthreads();
  PG_READ(0, 0, f0)

  // hipcc sinks an LDS read down to just before its first use; every PG_READ is therefore fenced with
  // sched_barrier so that the next group's operands are IN FLIGHT during the current group's MFMAs
  // (un-fenced, the four reads sat one MFMA ahead of their consumers and ~130 cycles of LDS latency were
  // exposed at each of the four group boundaries of a k-tile).
#define PG_FENCE __builtin_amdgcn_sched_barrier(0);
  int kt = 0;
  for (; kt + 1 < nk; ++kt) {
    const int cur = kt & 1;
    PG_LOAD(kt + 1)
    PG_READ(cur, 1, f1)
    PG_FENCE
    PG_MFMA(f0)                             // group 0
    PG_FENCE
    PG_READ(cur, 2, f0)
    PG_FENCE
    PG_MFMA(f1)                             // group 1
    PG_FENCE
    PG_STORE(cur ^ 1)                       // tile kt+1 -> spare buffer (last read two barriers ago)
    PG_READ(cur, 3, f1)
    PG_FENCE
    PG_MFMA(f0)                             // group 2
    PG_FENCE
    __syncthreads();                        // tile kt+1 visible; every wave has completed its reads of tile kt
    PG_READ(cur ^ 1, 0, f0)
    PG_FENCE
    PG_MFMA(f1)                             // group 3 of tile kt covers the first reads of tile kt+1
    PG_FENCE
  }
  {
    const int cur = kt & 1;
    PG_READ(cur, 1, f1)
    PG_FENCE
    PG_MFMA(f0)
    PG_FENCE
    PG_READ(cur, 2, f0)
    PG_FENCE
    PG_MFMA(f1)
    PG_FENCE
    PG_READ(cur, 3, f1)
    PG_FENCE
    PG_MFMA(f0)
    PG_FENCE
    PG_MFMA(f1)
  }

  // ---------------- epilogue ----------------
  const int mw = m0 + wm * 64;
  const int nw = n0 + wn * (32 * WNT);
  if (EPI == EPI_GEGLU) {
#pragma unroll
    for (int jp = 0; jp < WNT / 2; ++jp) {
      const int nout = ((nw + 64 * jp) >> 1) + l31;
      const float bh = p.bias ? p.bias[nw + 64 * jp + l31] : 0.f;
      const float bg = p.bias ? p.bias[nw + 64 * jp + 32 + l31] : 0.f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mw + mi * 32 + mfma32_crow(r, hi);
          if (m < p.M) {
            const float h = acc[mi][2 * jp][r] + bh;
            const float g = acc[mi][2 * jp + 1][r] + bg;
            const float ge = 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f));
            p.C[(size_t)m * p.ldc + nout] = h * ge;
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int ni = 0; ni < WNT; ++ni) {
      const int n = nw + ni * 32 + l31;
      const float bn = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mw + mi * 32 + mfma32_crow(r, hi);
        if (m >= p.M) continue;
        float v = acc[mi][ni][r] + bn;
        if (EPI == EPI_BIAS) {
          p.C[(size_t)m * p.ldc + n] = v;
        } else if (EPI == EPI_BIAS_RESID) {
          p.C[(size_t)m * p.ldc + n] = p.resid[(size_t)m * p.ldr + n] + v;
        } else if (EPI == EPI_BIAS_SILU) {
          p.C[(size_t)m * p.ldc + n] = v / (1.0f + expf(-v));
        } else if (EPI == EPI_BIAS_RELU) {
          p.C[(size_t)m * p.ldc + n] = fmaxf(v, 0.f);
        } else if (EPI == EPI_BIAS_ANCHOR) {
          const int sel = p.anchor[m] ? 1 : 0;
          p.C[(size_t)m * p.ldc + n] = v + p.anchor_emb[(size_t)sel * p.N + n];
        } else if (EPI == EPI_QKV_HEADMAJOR) {
          const int dmodel = p.heads * 64;
          const int c = n / dmodel;
          const int rem = n - c * dmodel;
          const int h = rem >> 6, j = rem & 63;
          p.C[(((size_t)c * p.heads + h) * p.M + m) * 64 + j] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant (the default).  Ablation on MI355X (scripts/mfma_ablation.py) shows what costs the
// register-staged loop its MFMA rate: a pure MFMA stream with LDS operand reads and a barrier per 64 MFMAs
// sustains ~141 TF, adding the 8 global_load_dwordx4 + 8 ds_write_b128 per wave per k-tile drops it to ~122 TF.
// Here the k-tiles go global -> LDS directly (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass,
// 32 fewer VGPRs).  The DMA writes LDS lane-linearly (wave base + lane*16 B), so rows cannot be padded;
// bank conflicts are avoided by an XOR swizzle of the 16-byte slot index, slot' = slot ^ ((row >> 1) & 7),
// applied to the per-lane GLOBAL source address when staging and to the ds_read_b128 address when reading
// (conflict-free for every 16-lane group of ds_read_b128: rows {0-3,12-15,20-27} x one logical slot map to
// 16 distinct physical slots of the 256-byte bank row).
// ---------------------------------------------------------------------------------------------
__device__ unsigned int g_gemm_cu_arrivals[16 * 256];
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_dma_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(1024))) float smem[2 * 2 * 128 * 32];   // 64 KB: [A0 A1 B0 B1][128 rows][32 floats]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  const int nt = p.N / GBN;
  const int mt = (p.M + GBM - 1) / GBM;
  const int logical = xcd_remap(blockIdx.x, mt * nt);
  const int m0 = (logical / nt) * GBM;
  const int n0 = (logical % nt) * GBN;

  // Two blocks share a CU (2 waves per SIMD) and all blocks take the same time, so blocks launched together stay IN PHASE for
  // the whole kernel: both wait for their first tile, both run their epilogue at the same moment and the matrix pipe idles.  The
  // second resident block of every CU therefore starts half a block-duration late (one block = K/32 k-tiles x 64 MFMAs x 64 cycles,
  // twice that with the SIMD shared), after which each block's prologue / epilogue falls into the other's main loop.
  if (p.stagger && blockIdx.x < 512) {
    bool late = blockIdx.x >= 256;                                       // 1: assume breadth-first placement of the first 512 blocks
    if (p.stagger == 2) {                                                // 2: ask the hardware which CU this is, count arrivals
      __shared__ unsigned int slot_s;
      if (tid == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u;
        slot_s = atomicAdd(&g_gemm_cu_arrivals[(xcc << 8) | ((hw >> 8) & 0xffu)], 1u);
      }
      __syncthreads();
      late = slot_s & 1u;
    }
    if (late) {
      const unsigned long long t0 = wall_clock64();                        // 100 MHz
      unsigned long long ticks = (unsigned long long)(p.K / GBK) * 178ull;         // 4096 cycles at ~2.3 GHz per k-tile
      ticks = ticks < 20000ull ? ticks : 20000ull;                                 // never more than 0.2 ms
      while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    }
  }

  // DMA sources: chunk id = i*256 + tid -> (row = id >> 3, physical slot = id & 7) holds logical slot (slot ^ swz(row))
  const float* a_src[4];
  const float* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = i * 256 + tid;
    const int row = id >> 3;
    const int lslot = (id & 7) ^ ((row >> 1) & 7);
    int r = m0 + row;
    r = r < p.M ? r : p.M - 1;
    a_src[i] = p.A + (size_t)r * p.lda + 4 * lslot;
    w_src[i] = p.W + (size_t)(n0 + row) * p.ldw + 4 * lslot;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  struct Frag { float4 a0, a1, b0, b1; } f0, f1;
  int nk = p.K / GBK;
  if (EPI == EPI_SPLITK_PART) {                      // this block's share of the k-tiles
    const int k0 = (int)((long)nk * blockIdx.y / gridDim.y), k1 = (int)((long)nk * (blockIdx.y + 1) / gridDim.y);
#pragma unroll
    for (int i = 0; i < 4; ++i) { a_src[i] += (size_t)k0 * GBK; w_src[i] += (size_t)k0 * GBK; }
    nk = k1 - k0;
  }
  const int sw = (l31 >> 1) & 7;
  const int a_row = (wm * 64 + l31) * 32;
  const int b_row = 2 * 4096 + (wn * 64 + l31) * 32;
  const int co0 = ((0 + hi) ^ sw) * 4, co1 = ((2 + hi) ^ sw) * 4, co2 = ((4 + hi) ^ sw) * 4, co3 = ((6 + hi) ^ sw) * 4;

  // The DMA is issued from inline asm: through the builtin, hipcc (ROCm 7.2) treats the transfer as an LDS write that
  // may alias every later ds_read and drains it (s_waitcnt vmcnt(0)) before the first fragment read of the SAME
  // iteration, which serialises the whole pipeline.  Hidden in asm, the transfer is retired by the explicit
  // vmcnt(0) in front of the barrier that publishes the tile (DG_SYNC).
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);   // + chunk bytes below
#define DG_DMA1(GSRC, LDSB)                                                                                   \
  {                                                                                                           \
    unsigned keep_;                                                                                           \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(GSRC), "s"(LDSB) : "memory");                                           \
  }
#define DG_DMA(KT, BUF)                                                                                       \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                             \
    DG_DMA1(a_src[i] + (size_t)(KT) * GBK, lds_wave + (unsigned)(((BUF) * 4096 + i * 1024) * 4))              \
    DG_DMA1(w_src[i] + (size_t)(KT) * GBK, lds_wave + (unsigned)((8192 + (BUF) * 4096 + i * 1024) * 4))       \
  }
#define DG_SYNC asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
#define DG_RD(OFF) (*reinterpret_cast<const float4*>(&smem[OFF]))
#define DG_READ(BUF, CO, F)                                   \
  F.a0 = DG_RD((BUF) * 4096 + a_row + (CO));                  \
  F.a1 = DG_RD((BUF) * 4096 + a_row + 32 * 32 + (CO));        \
  F.b0 = DG_RD((BUF) * 4096 + b_row + (CO));                  \
  F.b1 = DG_RD((BUF) * 4096 + b_row + 32 * 32 + (CO));
#define DG_MM(I, J, AV, BV) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV, BV, acc[I][J], 0, 0, 0);
#define DG_STEP(F, C) DG_MM(0, 0, F.a0.C, F.b0.C) DG_MM(0, 1, F.a0.C, F.b1.C) DG_MM(1, 0, F.a1.C, F.b0.C) DG_MM(1, 1, F.a1.C, F.b1.C)
#define DG_MFMA(F) DG_STEP(F, x) DG_STEP(F, y) DG_STEP(F, z) DG_STEP(F, w)
#define DG_FENCE __builtin_amdgcn_sched_barrier(0);

  DG_DMA(0, 0)
  DG_SYNC
  DG_READ(0, co0, f0)

  int kt = 0;
  for (; kt + 1 < nk; ++kt) {
    const int cur = kt & 1;
    DG_DMA(kt + 1, cur ^ 1)                 // spare buffer: every wave finished reading it before the last barrier
    DG_READ(cur, co1, f1)
    DG_FENCE
    DG_MFMA(f0)
    DG_FENCE
    DG_READ(cur, co2, f0)
    DG_FENCE
    DG_MFMA(f1)
    DG_FENCE
    DG_READ(cur, co3, f1)
    DG_FENCE
    DG_MFMA(f0)
    DG_FENCE
    DG_SYNC                                 // tile kt+1 landed (vmcnt(0)) and visible; reads of tile kt complete
    DG_READ(cur ^ 1, co0, f0)
    DG_FENCE
    DG_MFMA(f1)
    DG_FENCE
  }
  {
    const int cur = kt & 1;
    DG_READ(cur, co1, f1)
    DG_FENCE
    DG_MFMA(f0)
    DG_FENCE
    DG_READ(cur, co2, f0)
    DG_FENCE
    DG_MFMA(f1)
    DG_FENCE
    DG_READ(cur, co3, f1)
    DG_FENCE
    DG_MFMA(f0)
    DG_FENCE
    DG_MFMA(f1)
  }

  // ---------------- epilogue (same as the other variants) ----------------
  const int mw = m0 + wm * 64;
  const int nw = n0 + wn * 64;
  if (EPI == EPI_QKV_HEADMAJOR && p.gamma_q && nw < 2 * p.heads * 64) { qkv_store_normalised<2>(p, acc, mw, nw, hi, l31); return; }
  if (EPI == EPI_GEGLU) {
    const int nout = (nw >> 1) + l31;
    const float bh = p.bias ? p.bias[nw + l31] : 0.f;
    const float bg = p.bias ? p.bias[nw + 32 + l31] : 0.f;
    if (p.geglu_fast) {                              // tuning key 9: packed-pair GEGLU (see half.h), stage-major over 8 pairs
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        f32x2 h2[8], g2[8], o2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          h2[j] = f32x2{acc[mi][0][2 * j], acc[mi][0][2 * j + 1]} + f32x2{bh, bh};
          g2[j] = f32x2{acc[mi][1][2 * j], acc[mi][1][2 * j + 1]} + f32x2{bg, bg};
        }
        geglu_pairs<8>(h2, g2, o2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int m = mw + mi * 32 + mfma32_crow(2 * j, hi);
          if (m < p.M) p.C[(size_t)m * p.ldc + nout] = o2[j].x;
          if (m + 1 < p.M) p.C[(size_t)(m + 1) * p.ldc + nout] = o2[j].y;
        }
      }
      return;
    }
    auto geglu = [&](auto G) {                       // G: guard rows against M (only the last row tile needs it)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mw + mi * 32 + mfma32_crow(r, hi);
          if (decltype(G)::value && m >= p.M) continue;
          const float h = acc[mi][0][r] + bh;
          const float g = acc[mi][1][r] + bg;
          const float ge = 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f));
          p.C[(size_t)m * p.ldc + nout] = h * ge;
        }
      }
    };
    if (mw + 64 <= p.M) geglu(std::false_type{}); else geglu(std::true_type{});
    return;
  }
  if (EPI == EPI_BIAS_RESID) {
    // r02: the residual rows are loaded UNCONDITIONALLY (row index clamped) and all 64 loads of the wave tile are issued before the
    // first use -- the per-register `if (m < M)` of the generic loop made hipcc emit load -> wait -> add -> store 64 times in a row
    // (64 exposed memory latencies per tile: the K = 512 shapes ran at 0.67-0.70 of the matrix peak against 0.84 at K = 2048).
    const bool full = mw + 64 <= p.M;
    float rr[2][2][16];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int m = mw + mi * 32 + mfma32_crow(r, hi);
          m = m < p.M ? m : p.M - 1;
          rr[mi][ni][r] = p.resid[(size_t)m * p.ldr + nw + ni * 32 + l31];
        }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = nw + ni * 32 + l31;
        const float bn = p.bias ? p.bias[n] : 0.f;
        if (full) {                                  // wave-uniform: straight-line stores
#pragma unroll
          for (int r = 0; r < 16; ++r)
            p.C[(size_t)(mw + mi * 32 + mfma32_crow(r, hi)) * p.ldc + n] = rr[mi][ni][r] + (acc[mi][ni][r] + bn);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mw + mi * 32 + mfma32_crow(r, hi);
            if (m < p.M) p.C[(size_t)m * p.ldc + n] = rr[mi][ni][r] + (acc[mi][ni][r] + bn);
          }
        }
      }
    return;
  }
  auto store_tile = [&](auto G) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = nw + ni * 32 + l31;
        const float bn = p.bias ? p.bias[n] : 0.f;
        size_t qkv_col = 0;                            // head-major plane and column of this lane's output column (one division per column, not per element)
        if (EPI == EPI_QKV_HEADMAJOR) {
          const int dmodel = p.heads * 64;
          const int c = n / dmodel;
          const int rem = n - c * dmodel;
          qkv_col = ((size_t)c * p.heads + (rem >> 6)) * (size_t)p.M * 64 + (rem & 63);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mw + mi * 32 + mfma32_crow(r, hi);
          if (decltype(G)::value && m >= p.M) continue;
          float v = acc[mi][ni][r] + bn;
          if (EPI == EPI_BIAS) {
            p.C[(size_t)m * p.ldc + n] = v;
          } else if (EPI == EPI_SPLITK_PART) {
            p.splitk_ws[((size_t)blockIdx.y * p.M + m) * p.N + n] = acc[mi][ni][r];
          } else if (EPI == EPI_BIAS_SILU) {
            p.C[(size_t)m * p.ldc + n] = v / (1.0f + expf(-v));
          } else if (EPI == EPI_BIAS_RELU) {
            p.C[(size_t)m * p.ldc + n] = fmaxf(v, 0.f);
          } else if (EPI == EPI_BIAS_ANCHOR) {
            const int sel = p.anchor[m] ? 1 : 0;
            p.C[(size_t)m * p.ldc + n] = v + p.anchor_emb[(size_t)sel * p.N + n];
          } else if (EPI == EPI_QKV_HEADMAJOR) {
            p.C[qkv_col + (size_t)m * 64] = v;
          }
        }
      }
    }
  };
  if (mw + 64 <= p.M) store_tile(std::false_type{}); else store_tile(std::true_type{});
}

// ---------------------------------------------------------------------------------------------
// variant 32 (opt-in, rap_set_tuning(0, 32)): the same LDS-DMA loop on a 256x256 block tile, 8 waves (2 x 4), wave tile
// 128 x 64 = 4 x 2 MFMA tiles -- the shape of the 16-bit GEMM (gemm_h16.hip).  Per 4 k-values a wave reads 6 fragments for 32
// MFMAs (the 128x128 kernel: 4 for 16) and meets a barrier every 256 MFMAs instead of every 64; one block per CU (128 KB LDS), so
// prologue and epilogue are not covered by a second block.  N % 256 == 0, else the launcher falls back to the 128x128 kernel.
// ---------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512) void gemm_f32_dma256_kernel(GemmParams p) {
  constexpr int WN = 4, TM = 4, TN = 2, NT = 512, BM = 256, BN = 256;
  constexpr int CA = BM * 8 / NT, CB = BN * 8 / NT;          // 16-byte chunks per thread per k-tile
  constexpr int ABYTES = BM * 128, BBYTES = BN * 128;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem256[];   // [A0 A1 B0 B1]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int l31 = lane & 31;
  const int wm = wave / WN, wn = wave % WN;

  const int nt = p.N / BN;
  const int mt = (p.M + BM - 1) / BM;
  const int logical = xcd_remap(blockIdx.x, mt * nt);
  const int m0 = (logical / nt) * BM;
  const int n0 = (logical % nt) * BN;

  const float* a_src[CA];
  const float* w_src[CB];
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int id = i * NT + tid;
    const int row = id >> 3;
    const int lslot = (id & 7) ^ ((row >> 1) & 7);
    int r = m0 + row;
    r = r < p.M ? r : p.M - 1;
    a_src[i] = p.A + (size_t)r * p.lda + 4 * lslot;
  }
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    const int id = i * NT + tid;
    const int row = id >> 3;
    const int lslot = (id & 7) ^ ((row >> 1) & 7);
    w_src[i] = p.W + (size_t)(n0 + row) * p.ldw + 4 * lslot;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / GBK;
  const int sw = (l31 >> 1) & 7;
  const int a_row = (wm * TM * 32 + l31) * 128;              // byte offsets inside one A / B buffer
  const int b_row = (wn * TN * 32 + l31) * 128;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem256;
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
#define BG_DMA1(GSRC, LDSB)                                                                                   \
  {                                                                                                           \
    unsigned keep_;                                                                                           \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(GSRC), "s"(LDSB) : "memory");                                           \
  }
#define BG_DMA(KT, BUF)                                                                                       \
  _Pragma("unroll") for (int i = 0; i < CA; ++i)                                                              \
    BG_DMA1(a_src[i] + (size_t)(KT) * GBK, lds_wave + (unsigned)((BUF) * ABYTES + i * NT * 16))               \
  _Pragma("unroll") for (int i = 0; i < CB; ++i)                                                              \
    BG_DMA1(w_src[i] + (size_t)(KT) * GBK, lds_wave + (unsigned)(2 * ABYTES + (BUF) * BBYTES + i * NT * 16))
#define BG_SYNC asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
#define BG_FENCE __builtin_amdgcn_sched_barrier(0);

  struct Frag { float4 a[TM]; float4 b[TN]; };
  Frag f0, f1;
  auto read_frag = [&](Frag& f, int buf, int g) {
    const int co = ((2 * g + hi) ^ sw) * 16;
#pragma unroll
    for (int i = 0; i < TM; ++i)
      f.a[i] = *reinterpret_cast<const float4*>(smem256 + buf * ABYTES + a_row + i * 32 * 128 + co);
#pragma unroll
    for (int j = 0; j < TN; ++j)
      f.b[j] = *reinterpret_cast<const float4*>(smem256 + 2 * ABYTES + buf * BBYTES + b_row + j * 32 * 128 + co);
  };
#define BG_MMA_C(F, C)                                                                                  \
  _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                        \
    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                      \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a[i].C, F.b[j].C, acc[i][j], 0, 0, 0);
#define BG_MMA(F) BG_MMA_C(F, x) BG_MMA_C(F, y) BG_MMA_C(F, z) BG_MMA_C(F, w)

  BG_DMA(0, 0)
  BG_SYNC
  read_frag(f0, 0, 0);

  int kt = 0;
  for (; kt + 1 < nk; ++kt) {
    const int cur = kt & 1;
    BG_DMA(kt + 1, cur ^ 1)
    read_frag(f1, cur, 1);
    BG_FENCE
    BG_MMA(f0)
    BG_FENCE
    read_frag(f0, cur, 2);
    BG_FENCE
    BG_MMA(f1)
    BG_FENCE
    read_frag(f1, cur, 3);
    BG_FENCE
    BG_MMA(f0)
    BG_FENCE
    BG_SYNC
    read_frag(f0, cur ^ 1, 0);
    BG_FENCE
    BG_MMA(f1)
    BG_FENCE
  }
  {
    const int cur = kt & 1;
    read_frag(f1, cur, 1);
    BG_FENCE
    BG_MMA(f0)
    BG_FENCE
    read_frag(f0, cur, 2);
    BG_FENCE
    BG_MMA(f1)
    BG_FENCE
    read_frag(f1, cur, 3);
    BG_FENCE
    BG_MMA(f0)
    BG_FENCE
    BG_MMA(f1)
  }

  // ---------------- epilogue: as the 128x128 kernels, over 4 x 2 MFMA tiles ----------------
  const int mw = m0 + wm * TM * 32;
  const int nw = n0 + wn * TN * 32;
  if (EPI == EPI_QKV_HEADMAJOR && p.gamma_q && nw < 2 * p.heads * 64) { qkv_store_normalised<TM>(p, acc, mw, nw, hi, l31); return; }
  if (EPI == EPI_GEGLU) {
    const int nout = (nw >> 1) + l31;
    const float bh = p.bias ? p.bias[nw + l31] : 0.f;
    const float bg = p.bias ? p.bias[nw + 32 + l31] : 0.f;
    if (p.geglu_fast) {                              // tuning key 9: packed-pair GEGLU (see half.h), stage-major over 8 pairs
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        f32x2 h2[8], g2[8], o2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          h2[j] = f32x2{acc[mi][0][2 * j], acc[mi][0][2 * j + 1]} + f32x2{bh, bh};
          g2[j] = f32x2{acc[mi][1][2 * j], acc[mi][1][2 * j + 1]} + f32x2{bg, bg};
        }
        geglu_pairs<8>(h2, g2, o2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int m = mw + mi * 32 + mfma32_crow(2 * j, hi);
          if (m < p.M) p.C[(size_t)m * p.ldc + nout] = o2[j].x;
          if (m + 1 < p.M) p.C[(size_t)(m + 1) * p.ldc + nout] = o2[j].y;
        }
      }
      return;
    }
    auto geglu = [&](auto G) {
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mw + mi * 32 + mfma32_crow(r, hi);
          if (decltype(G)::value && m >= p.M) continue;
          const float h = acc[mi][0][r] + bh;
          const float g = acc[mi][1][r] + bg;
          const float ge = 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f));
          p.C[(size_t)m * p.ldc + nout] = h * ge;
        }
      }
    };
    if (mw + 32 * TM <= p.M) geglu(std::false_type{}); else geglu(std::true_type{});
    return;
  }
  if (EPI == EPI_BIAS_RESID) {                     // see gemm_f32_dma_kernel: unconditional, batched residual loads
    const bool full = mw + 32 * TM <= p.M;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
      float rr[TN][16];
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int m = mw + mi * 32 + mfma32_crow(r, hi);
          m = m < p.M ? m : p.M - 1;
          rr[ni][r] = p.resid[(size_t)m * p.ldr + nw + ni * 32 + l31];
        }
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const int n = nw + ni * 32 + l31;
        const float bn = p.bias ? p.bias[n] : 0.f;
        if (full) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            p.C[(size_t)(mw + mi * 32 + mfma32_crow(r, hi)) * p.ldc + n] = rr[ni][r] + (acc[mi][ni][r] + bn);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mw + mi * 32 + mfma32_crow(r, hi);
            if (m < p.M) p.C[(size_t)m * p.ldc + n] = rr[ni][r] + (acc[mi][ni][r] + bn);
          }
        }
      }
    }
    return;
  }
  auto store_tile = [&](auto G) {
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const int n = nw + ni * 32 + l31;
      const float bn = p.bias ? p.bias[n] : 0.f;
      size_t qkv_col = 0;                              // head-major plane and column of this lane's output column
      if (EPI == EPI_QKV_HEADMAJOR) {
        const int dmodel = p.heads * 64;
        const int c = n / dmodel;
        const int rem = n - c * dmodel;
        qkv_col = ((size_t)c * p.heads + (rem >> 6)) * (size_t)p.M * 64 + (rem & 63);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mw + mi * 32 + mfma32_crow(r, hi);
        if (decltype(G)::value && m >= p.M) continue;
        const float v = acc[mi][ni][r] + bn;
        if (EPI == EPI_BIAS) {
          p.C[(size_t)m * p.ldc + n] = v;
        } else if (EPI == EPI_BIAS_SILU) {
          p.C[(size_t)m * p.ldc + n] = v / (1.0f + expf(-v));
        } else if (EPI == EPI_BIAS_RELU) {
          p.C[(size_t)m * p.ldc + n] = fmaxf(v, 0.f);
        } else if (EPI == EPI_BIAS_ANCHOR) {
          const int sel = p.anchor[m] ? 1 : 0;
          p.C[(size_t)m * p.ldc + n] = v + p.anchor_emb[(size_t)sel * p.N + n];
        } else if (EPI == EPI_QKV_HEADMAJOR) {
          p.C[qkv_col + (size_t)m * 64] = v;
        }
      }
    }
  }
  };
  if (mw + 32 * TM <= p.M) store_tile(std::false_type{}); else store_tile(std::true_type{});
}

// C[m][n] = resid[m][n] + bias[n] + sum_s part[s][m][n]   (the split-K path of EPI_BIAS_RESID; one thread per 4 columns)
__global__ __launch_bounds__(256) void gemm_splitk_combine_kernel(GemmParams p, int splits) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = p.N / 4;
  if (i >= (long)p.M * n4) return;
  const long m = i / n4;
  const int n = (int)(i % n4) * 4;
  float4 acc = *reinterpret_cast<const float4*>(p.resid + m * p.ldr + n);
  if (p.bias) {
    const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
    acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
  }
  for (int s = 0; s < splits; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(p.splitk_ws + ((size_t)s * p.M + m) * p.N + n);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  *reinterpret_cast<float4*>(p.C + m * p.ldc + n) = acc;
}

// tuning knob (rap_set_tuning key 0): 0 = v1 128x128, 2 = pipelined 128x128 (two 4-wave blocks per CU),
// 4 = pipelined 128x256 (one 4-wave block per CU; N % 256 == 0, else falls back to 2),
// 8 = pipelined 256x128, one 8-wave block per CU with a static priority split per SIMD pair,
// 16 = LDS-DMA staged 128x128, 32 = LDS-DMA staged 256x256 (8 waves, one block per CU),
// 48 = per shape (default, r02): the 256x256 kernel where r01 run 54 measured it faster -- wide outputs at K <= 512 (qkv 111 -> 115 TF,
// ff1 123 -> 126) -- and the 128x128 kernel elsewhere (out-projection 105 vs 91, ff2 132 vs 127, embedding, head).
rap_tuning_t g_rap_gemm_variant = 48;

template <int EPI>
static void launch_gemm_variant(hipStream_t stream, const GemmParams& p, int variant) {
  const int mt = (p.M + GBM - 1) / GBM;
  if (variant == 32 && p.N % 256 == 0) {
    constexpr int LDS = 2 * (256 + 256) * 128;
    static bool attr_done = false;
    auto kern = gemm_f32_dma256_kernel<EPI>;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(((p.M + 255) / 256) * (p.N / 256)), dim3(512), LDS, stream, p);
  } else if (variant == 16 || variant == 32) {
    hipLaunchKernelGGL(gemm_f32_dma_kernel<EPI>, dim3(mt * (p.N / GBN)), dim3(256), 0, stream, p);
  } else if (variant == 8) {
    hipLaunchKernelGGL((gemm_f32_pipe_kernel<EPI, 2, 4>), dim3(((p.M + 255) / 256) * (p.N / GBN)), dim3(512), 0, stream, p);
  } else if (variant == 4 && p.N % 256 == 0) {
    hipLaunchKernelGGL((gemm_f32_pipe_kernel<EPI, 4, 2>), dim3(mt * (p.N / 256)), dim3(256), 0, stream, p);
  } else if (variant == 0) {
    hipLaunchKernelGGL(gemm_f32_kernel<EPI>, dim3(mt * (p.N / GBN)), dim3(256), 0, stream, p);
  } else {
    hipLaunchKernelGGL((gemm_f32_pipe_kernel<EPI, 2, 2>), dim3(mt * (p.N / GBN)), dim3(256), 0, stream, p);
  }
}

rap_tuning_t g_rap_gemm_splitk = 1;      // tuning key 6: 0 = never split K for few-row calls
rap_tuning_t g_rap_gemm_stagger = 1;     // measured (r01 run 39): +1.3 % on the K = 512 shapes, +1.2 % at K = 2048; 2 (by CU id) is no better
// tuning key 9 (fp32 path): GEGLU's Phi(g).  1 (default since r02 call 50) = Abramowitz-Stegun 7.1.26 erfc, |error| <= 1.5e-7 absolute
// (about one fp32 ulp of Phi near 1/2), on the packed fp32 pipe, stage-major over eight output pairs (half.h: geglu_pairs) -- the
// epilogue of the largest GEMM of a layer was ~50 scalar VALU instructions per output with erff: ff1 8.35 -> 8.09 ms (131.7 -> 135.9 TF),
// headline 15 940 -> 16 049 points/s, all-step deviation from the unmodified reference unchanged (final cloud 5.4e-7 vs 6.6e-7,
// per-step maximum 9.5e-7 vs 8.3e-7: the fp32 noise floor).  0 = libm-grade erff.
rap_tuning_t g_rap_geglu_fast = 1;
int launch_gemm_f32(hipStream_t stream, int epilogue, const GemmParams& p_in) {
  GemmParams p = p_in;
  p.geglu_fast = g_rap_geglu_fast;
  p.stagger = ((long)((p.M + GBM - 1) / GBM) * (p.N / GBN) >= 1024) ? g_rap_gemm_stagger.load() : 0;     // only when the chip is filled twice over
  if (p.M <= 0) return RAP_OK;
  if (p.N % GBN != 0 || p.K % GBK != 0 || p.K <= 0) return RAP_ERR_INVALID;
  if ((p.lda & 3) || (p.ldw & 3)) return RAP_ERR_INVALID;
  int v = g_rap_gemm_variant;
  // per shape (r02 call 40, after the epilogue rewrite): the 256x256 kernel wherever there are at least two rounds of its tiles and
  // K >= 256 -- qkv 128 vs 119 TF, ff1 132 vs 129, ff2 139 vs 131, head 125 vs 119, out-projection equal; the K = 64 embedding GEMM and
  // few-token calls stay on the 128x128 kernel (two blocks per CU cover each other's prologue).
  if (v == 48) v = (p.N % 256 == 0 && p.K >= 256 && (long)((p.M + 255) / 256) * (p.N / 256) >= 512) ? 32 : 16;
  if (epilogue == EPI_BIAS_RESID && (v == 16 || v == 32) && p.splitk_ws && g_rap_gemm_splitk && p.K >= 1024 && (p.ldr & 3) == 0 && (p.ldc & 3) == 0 &&
      (long)((p.M + GBM - 1) / GBM) * (p.N / GBN) <= 128) {
    const int splits = 4;
    hipLaunchKernelGGL(gemm_f32_dma_kernel<EPI_SPLITK_PART>, dim3(((p.M + GBM - 1) / GBM) * (p.N / GBN), splits), dim3(256), 0, stream, p);
    RAP_LAUNCH_CHECK();
    hipLaunchKernelGGL(gemm_splitk_combine_kernel, dim3((unsigned)(((long)p.M * (p.N / 4) + 255) / 256)), dim3(256), 0, stream, p, splits);
    RAP_LAUNCH_CHECK();
    return RAP_OK;
  }
  switch (epilogue) {
    case EPI_BIAS: launch_gemm_variant<EPI_BIAS>(stream, p, v); break;
    case EPI_BIAS_RESID: launch_gemm_variant<EPI_BIAS_RESID>(stream, p, v); break;
    case EPI_BIAS_SILU: launch_gemm_variant<EPI_BIAS_SILU>(stream, p, v); break;
    case EPI_GEGLU: launch_gemm_variant<EPI_GEGLU>(stream, p, v); break;
    case EPI_QKV_HEADMAJOR:
      if (p.N != 3 * p.heads * 64) return RAP_ERR_INVALID;
      if (p.gamma_q && (!p.gamma_k || !(v == 16 || v == 32))) return RAP_ERR_INVALID;      // fused qk-norm lives in the LDS-DMA kernels
      launch_gemm_variant<EPI_QKV_HEADMAJOR>(stream, p, v);
      break;
    case EPI_BIAS_ANCHOR: launch_gemm_variant<EPI_BIAS_ANCHOR>(stream, p, v); break;
    case EPI_BIAS_RELU: launch_gemm_variant<EPI_BIAS_RELU>(stream, p, v); break;
    default: return RAP_ERR_INVALID;
  }
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
