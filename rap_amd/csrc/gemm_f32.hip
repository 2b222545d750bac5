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
//  * k-tiles go global -> LDS directly (global_load_lds_dwordx4, XOR slot swizzle instead of row padding), double-buffered
//    LDS, one barrier per k-tile.  (Rounds 1-2 also carried a register-staged v1 kernel and three software-pipelined
//    register-staged tilings, 12 points slower -- DESIGN.md 4.2; they are in the history at 72efb73.)
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

// Epilogue of one 128 x 64 wave tile (4 x 2 MFMA tiles) of the 256 x 256 kernels: as the 128x128 kernels' epilogues.
template <int EPI>
__device__ __forceinline__ void gemm_f32_epilogue_4x2(const GemmParams& p, f32x16 (&acc)[4][2], int mw, int nw, int hi, int l31) {
  constexpr int TM = 4, TN = 2;
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

  gemm_f32_epilogue_4x2<EPI>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, hi, l31);
}

// ---------------------------------------------------------------------------------------------
// Persistent form of the 256 x 256 kernel (round 3; the same idea as gemm_h16_php_kernel, where scripts/gemm_ts.py itemised the per-tile
// overhead of a one-tile-per-block launch: block start-up and tile decode, the wait for the first k-tile, the gap to the next block).
// ONE block per CU walks the tiles of its XCD (virtual block id v = blockIdx.x + i * gridDim.x, gridDim.x a multiple of 8) and the
// k-tiles of consecutive output tiles form one DMA stream: the last k-tile of an output tile requests the first k-tile of the NEXT one
// into the spare stage exactly as it would request its own successor, so the next k-loop starts on landed data and the epilogue's
// stores (straight from the accumulator registers: no LDS involved) drain under it.  Sources are a scalar base (tile origin + k offset)
// plus one 32-bit per-thread byte offset per 16-byte chunk (the saddr form of global_load_lds): needs M % 256 == 0 (no row clamping).
// Same MFMA order, same epilogue: results are bit-identical to gemm_f32_dma256_kernel.
// ---------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512) void gemm_f32_dma256p_kernel(GemmParams p) {
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
  const int total = (p.M / BM) * nt;
  const int nk = p.K / GBK;

  unsigned a_off[CA], w_off[CB];           // byte offsets inside the tile's A / W panel at k = 0
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int id = i * NT + tid;
    const int row = id >> 3;
    const int lslot = (id & 7) ^ ((row >> 1) & 7);
    a_off[i] = (unsigned)(row * p.lda + 4 * lslot) * 4u;
  }
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    const int id = i * NT + tid;
    const int row = id >> 3;
    const int lslot = (id & 7) ^ ((row >> 1) & 7);
    w_off[i] = (unsigned)(row * p.ldw + 4 * lslot) * 4u;
  }

  f32x16 acc[TM][TN];
  const int sw = (l31 >> 1) & 7;
  const int a_row = (wm * TM * 32 + l31) * 128;              // byte offsets inside one A / B buffer
  const int b_row = (wn * TN * 32 + l31) * 128;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem256;
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
#define BP_DMA1(VOFF, SBASE, LDSB)                                                                            \
  {                                                                                                           \
    unsigned keep_;                                                                                           \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(VOFF), "s"(LDSB), "s"(SBASE) : "memory");                                \
  }
#define BP_DMA(ABASE, WBASE, BUF)                                                                             \
  _Pragma("unroll") for (int i = 0; i < CA; ++i)                                                              \
    BP_DMA1(a_off[i], ABASE, lds_wave + (unsigned)((BUF) * ABYTES + i * NT * 16))                             \
  _Pragma("unroll") for (int i = 0; i < CB; ++i)                                                              \
    BP_DMA1(w_off[i], WBASE, lds_wave + (unsigned)(2 * ABYTES + (BUF) * BBYTES + i * NT * 16))
  // raw barrier (no memory fence needed: LDS-DMA arrival is confirmed by this wave's vmcnt, fragment reads of the stage that the next
  // DMA overwrites by lgkmcnt -- they were issued one MFMA group earlier, so the wait is free)
#define BP_SYNC asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);

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

  auto tile_bases = [&](int v, const unsigned char*& ab, const unsigned char*& wb, int& m0, int& n0) {
    const int logical = xcd_remap(v, total);
    m0 = (logical / nt) * BM;
    n0 = (logical % nt) * BN;
    ab = reinterpret_cast<const unsigned char*>(p.A) + (size_t)m0 * p.lda * 4;
    wb = reinterpret_cast<const unsigned char*>(p.W) + (size_t)n0 * p.ldw * 4;
  };

  int v = blockIdx.x;
  const unsigned char *a_cur, *w_cur, *a_nxt = nullptr, *w_nxt = nullptr;
  int m0, n0, m0n = 0, n0n = 0;
  tile_bases(v, a_cur, w_cur, m0, n0);
  BP_DMA(a_cur, w_cur, 0)
  BP_SYNC
  int par = 0;                              // stage of the current k-tile of the stream

  for (;;) {
    const int vn = v + (int)gridDim.x;
    const bool has_next = vn < total;
    if (has_next) tile_bases(vn, a_nxt, w_nxt, m0n, n0n);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    read_frag(f0, par, 0);

    for (int kt = 0; kt < nk; ++kt) {
      const int cur = par;
      const bool in1 = kt + 1 < nk;
      if (in1) { BP_DMA(a_cur + (size_t)(kt + 1) * (GBK * 4), w_cur + (size_t)(kt + 1) * (GBK * 4), cur ^ 1) }
      else if (has_next) { BP_DMA(a_nxt, w_nxt, cur ^ 1) }       // the stream continues into the next output tile
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
      BP_SYNC                              // successor k-tile landed (vmcnt(0): also this block's earlier stores) and visible; reads of `cur` complete
      if (in1) read_frag(f0, cur ^ 1, 0);
      BG_FENCE
      BG_MMA(f1)
      BG_FENCE
      par ^= 1;
    }
    gemm_f32_epilogue_4x2<EPI>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, hi, l31);
    if (!has_next) break;
    v = vn; a_cur = a_nxt; w_cur = w_nxt; m0 = m0n; n0 = n0n;
  }
}

// C[m][n] = resid[m][n] + bias[n] + sum_s part[s][m][n]   (the split-K path of EPI_BIAS_RESID; one thread per 4 columns)
// SILU (round 6, the two hidden layers of final_mlp in few-token calls): no residual, C = silu(bias + sum of the partials)
template <bool SILU>
__global__ __launch_bounds__(256) void gemm_splitk_combine_kernel(GemmParams p, int splits) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = p.N / 4;
  if (i >= (long)p.M * n4) return;
  const long m = i / n4;
  const int n = (int)(i % n4) * 4;
  float4 acc = SILU ? float4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const float4*>(p.resid + m * p.ldr + n);
  if (p.bias) {
    const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
    acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
  }
  for (int s = 0; s < splits; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(p.splitk_ws + ((size_t)s * p.M + m) * p.N + n);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (SILU) {
    acc.x = acc.x / (1.0f + expf(-acc.x)); acc.y = acc.y / (1.0f + expf(-acc.y));
    acc.z = acc.z / (1.0f + expf(-acc.z)); acc.w = acc.w / (1.0f + expf(-acc.w));
  }
  *reinterpret_cast<float4*>(p.C + m * p.ldc + n) = acc;
}

// Kernel choice: the LDS-DMA 256 x 256 / 8-wave kernel wherever there are at least two rounds of its tiles, N % 256 == 0 and
// K >= 256 (r02 call 40: qkv 128 vs 119 TF, ff1 132 vs 129, ff2 139 vs 131, head 125 vs 119, out-projection equal); the LDS-DMA
// 128 x 128 kernel (two blocks per CU cover each other's prologue) for the K = 64 embedding GEMM, narrow outputs and few-token calls.
// RAP_ABLATION_BUILD only: rap_set_tuning(0, 16 | 32) forces one of the two.
rap_tuning_t g_rap_gemm_variant = 48;

rap_tuning_t g_rap_gemm_f32_persistent = 1;     // tuning key 12: the persistent 256 x 256 kernel for full-tile shapes (1, default) or one tile per block (0)

template <int EPI>
static int launch_gemm_variant(hipStream_t stream, const GemmParams& p, int variant) {
  const int mt = (p.M + GBM - 1) / GBM;
  if (variant == 32 && p.N % 256 == 0 && g_rap_gemm_f32_persistent && p.M % 256 == 0 &&
      (long)(p.M / 256) * (p.N / 256) >= 512 && p.lda <= (1 << 20) && p.ldw <= (1 << 20)) {
    constexpr int LDS = 2 * (256 + 256) * 128;
    auto kern = gemm_f32_dma256p_kernel<EPI>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      rap_set_last_hip_error((int)hipGetLastError());
      return RAP_ERR_HIP;
    }
    static std::atomic<int> n_cu_cache[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RAP_ERR_HIP;
    int n_cu = (dev >= 0 && dev < 16) ? n_cu_cache[dev].load() : 0;
    if (n_cu == 0) {
      if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) return RAP_ERR_HIP;
      n_cu = n_cu >= 8 ? (n_cu / 8) * 8 : n_cu;
      if (dev >= 0 && dev < 16) n_cu_cache[dev] = n_cu;
    }
    const int total = (p.M / 256) * (p.N / 256);
    hipLaunchKernelGGL(kern, dim3(total < n_cu ? total : n_cu), dim3(512), LDS, stream, p);
  } else if (variant == 32 && p.N % 256 == 0) {
    constexpr int LDS = 2 * (256 + 256) * 128;
    auto kern = gemm_f32_dma256_kernel<EPI>;
    // per device and cheap: set unconditionally (a process may drive several GPUs)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      rap_set_last_hip_error((int)hipGetLastError());
      return RAP_ERR_HIP;
    }
    hipLaunchKernelGGL(kern, dim3(((p.M + 255) / 256) * (p.N / 256)), dim3(512), LDS, stream, p);
  } else {
    hipLaunchKernelGGL(gemm_f32_dma_kernel<EPI>, dim3(mt * (p.N / GBN)), dim3(256), 0, stream, p);
  }
  RAP_LAUNCH_CHECK();
  return RAP_OK;
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
  int v = 48;
#ifdef RAP_ABLATION_BUILD
  v = g_rap_gemm_variant;
#endif
  if (v != 16 && v != 32) v = (p.N % 256 == 0 && p.K >= 256 && (long)((p.M + 255) / 256) * (p.N / 256) >= 512) ? 32 : 16;
  // few-row calls: K >= 1024 (ff2) when the tile grid covers at most half of the CUs, K >= 512 (the out-projection) when it covers at
  // most a quarter (r03: one pair of 2 x 1024 points -- 64 tiles, a chain of 16 k-tiles each)
  const long tiles128 = (long)((p.M + GBM - 1) / GBM) * (p.N / GBN);
  if (epilogue == EPI_BIAS_RESID && (v == 16 || v == 32) && p.splitk_ws && g_rap_gemm_splitk && (p.ldr & 3) == 0 && (p.ldc & 3) == 0 &&
      ((p.K >= 1024 && tiles128 <= 128) || (p.K >= 512 && tiles128 <= 64))) {
    const int splits = 4;
    hipLaunchKernelGGL(gemm_f32_dma_kernel<EPI_SPLITK_PART>, dim3(((p.M + GBM - 1) / GBM) * (p.N / GBN), splits), dim3(256), 0, stream, p);
    RAP_LAUNCH_CHECK();
    hipLaunchKernelGGL(gemm_splitk_combine_kernel<false>, dim3((unsigned)(((long)p.M * (p.N / 4) + 255) / 256)), dim3(256), 0, stream, p, splits);
    RAP_LAUNCH_CHECK();
    return RAP_OK;
  }
  // the hidden layers of final_mlp (Linear + SiLU, fp32 in every mode) of a few-token call: 64 / 32 tiles of a 16-k-tile chain on 256 CUs
  // (r06 call 2: 39 us per launch, a fifth of a demo-size pair's per-step overhead) -- K over `splitk_planes` (2 or 4) blocks per tile
  if (epilogue == EPI_BIAS_SILU && p.splitk_ws && (p.splitk_planes == 2 || p.splitk_planes == 4) && g_rap_gemm_splitk && (p.ldc & 3) == 0 &&
      p.K >= 512 && (p.K / GBK) % p.splitk_planes == 0 && tiles128 <= 64) {
    hipLaunchKernelGGL(gemm_f32_dma_kernel<EPI_SPLITK_PART>, dim3(((p.M + GBM - 1) / GBM) * (p.N / GBN), p.splitk_planes), dim3(256), 0, stream, p);
    RAP_LAUNCH_CHECK();
    hipLaunchKernelGGL(gemm_splitk_combine_kernel<true>, dim3((unsigned)(((long)p.M * (p.N / 4) + 255) / 256)), dim3(256), 0, stream, p, p.splitk_planes);
    RAP_LAUNCH_CHECK();
    return RAP_OK;
  }
  switch (epilogue) {
    case EPI_BIAS: return launch_gemm_variant<EPI_BIAS>(stream, p, v);
    case EPI_BIAS_RESID: return launch_gemm_variant<EPI_BIAS_RESID>(stream, p, v);
    case EPI_BIAS_SILU: return launch_gemm_variant<EPI_BIAS_SILU>(stream, p, v);
    case EPI_GEGLU: return launch_gemm_variant<EPI_GEGLU>(stream, p, v);
    case EPI_QKV_HEADMAJOR:
      if (p.N != 3 * p.heads * 64) return RAP_ERR_INVALID;
      if (p.gamma_q && !p.gamma_k) return RAP_ERR_INVALID;
      return launch_gemm_variant<EPI_QKV_HEADMAJOR>(stream, p, v);
    case EPI_BIAS_ANCHOR: return launch_gemm_variant<EPI_BIAS_ANCHOR>(stream, p, v);
    case EPI_BIAS_RELU: return launch_gemm_variant<EPI_BIAS_RELU>(stream, p, v);
    default: return RAP_ERR_INVALID;
  }
}
