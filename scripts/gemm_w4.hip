// Prototype (round 3, NOT part of librapflow): bf16 GEMM C (M,N) = A (M,K) W (N,K)^T with FOUR waves per 256 x 256 block tile, i.e. a
// 128 x 128 tile per wave and ONE wave per SIMD (512 registers: 256 accumulators + fragments), against the shipped kernel's eight waves of
// 128 x 64 (two waves per SIMD, phase-split, 8 barriers per k-tile).  Question it answers: does register blocking (one third fewer LDS
// bytes per MFMA: 32 ds_read_b128 per 64 MFMAs instead of 28 per 32) and one barrier per k-tile beat the two-waves-per-SIMD hand-off
// at the layer shapes (K = 512 / 2048) and on the guide's square shape?  MI355X_MICROARCH.md prices a one-wave-per-SIMD stream at 32.4
// (hand-placed) to 35.8 (compiler-scheduled) cycles per MFMA with <= 5 fillers per gap.
//   * tiles by LDS-DMA (global_load_lds_dwordx4, scalar base + 32-bit per-thread offset), two 64 KB stages, 128-byte rows with the
//     slot ^ ((row >> 1) & 7) swizzle; 16 pieces per wave per k-tile, four per k-step, issued between the MFMAs of that step;
//   * swapped product (W fragment as the A operand): a lane owns a row of C, results leave as 8-byte pieces (MODE 0) or not at all
//     (MODE 1: main loop only; the store is guarded by a condition that is never true);
//   * one block per CU walks its XCD's tiles (persistent, as the shipped kernel) -- MODE bit 1 clear: one tile per block;
//   * MODE bit 4: epilogue through a per-wave LDS slab (ds_write_b64 of the lane's 4-column groups, whole 256-byte rows out);
//   * ablation of the main loop (with MODE bits 0 and 1 set): bit 2 = no LDS-DMA after the prologue, bit 3 = no fragment reads after the
//     first (the MFMAs run on stale registers): 3 = full loop, 7 = MFMA + reads, 11 = MFMA + DMA, 15 = MFMA + barrier only.
//   * MODE bit 6 (round 6): the W operand does NOT go through LDS.  The weights are static, so they are packed once in
//     MFMA-fragment order (pack_w_kernel: [32-column block][k-tile][k-step][lane][8 bf16] = one contiguous KB per fragment load) and every
//     wave loads its own fragments global -> VGPR (global_load_dwordx4, scalar base + lane * 16) one k-tile ahead into a second register
//     set; the LDS-DMA stream carries A only (8 pieces per k-tile instead of 16) and the four W fragment reads per k-step disappear.  The
//     price: the two wave rows of a 256 x 256 tile multiply the same 256 weight columns, LDS served them once, registers serve each its own
//     copy (64 KB of W per k-tile per CU from L2 instead of 32).
// extern "C" double gemm_w4(int M, int N, int K, int mode, int iters, double* max_err)  -> TFLOP/s (max_err vs a naive kernel on a sample of C)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <type_traits>
#include <vector>

typedef unsigned short u16;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// m0 is declared clobbered instead of saved and restored around every piece (two SALU fewer per MFMA gap)
#define W4_DMA1(VOFF, SBASE, LDSB)                                                                            \
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(VOFF), "s"(LDSB), "s"(SBASE) : "memory", "m0");
#define W4_FENCE __builtin_amdgcn_sched_barrier(0);

// one W fragment (8 bf16 per lane, 1 KB per wave) global -> VGPR: scalar base + per-lane byte offset; the caller counts it on vmcnt
#define W4_GLD(DST, VOFF, SBASE) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(DST) : "v"(VOFF), "s"(SBASE) : "memory");

template <int MODE>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(const u16* __restrict__ A, int lda, const u16* __restrict__ W, int ldw,
                                                         u16* __restrict__ C, int ldc, int M, int N, int K, const u16* __restrict__ Wp) {
  constexpr bool BREG = (MODE & 64) != 0;
  constexpr int STAGE = 65536, WOFF = 32768;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int wr = wave >> 1, wc = wave & 1;
  const int nt = N / 256;
  const int total = (M / 256) * nt;
  const int nk = K / 64;

  // DMA: piece i (0..7) of an operand = LDS rows 32 i + 8 wave + lane / 8 (1 KB, lane-linear), physical slot lane % 8
  const int r0 = wave * 8 + (lane >> 3);
  const int lslot = (lane & 7) ^ ((r0 >> 1) & 7);            // (row >> 1) & 7 is the same for rows r0 + 32 i
  const unsigned voffA = (unsigned)(r0 * lda + 8 * lslot) * 2u;
  const unsigned voffW = (unsigned)(r0 * ldw + 8 * lslot) * 2u;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
  const size_t a_step = (size_t)32 * lda * 2, w_step = (size_t)32 * ldw * 2;     // bytes between the rows of consecutive pieces

  const int sw = (l31 >> 1) & 7;
  const unsigned char* a_rd = smem + (wr * 128 + l31) * 128;
  const unsigned char* w_rd = smem + WOFF + (wc * 128 + l31) * 128;

  f32x16 acc[4][4];
  uint4 fa[2][4], fw[2][4];
  auto read_frags = [&](int set, int stage, int ks) {
    const int co = ((2 * ks + hi) ^ sw) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[set][i] = *reinterpret_cast<const uint4*>(a_rd + stage * STAGE + i * 4096 + co);
#pragma unroll
    for (int j = 0; j < 4; ++j) fw[set][j] = *reinterpret_cast<const uint4*>(w_rd + stage * STAGE + j * 4096 + co);
  };
  // swapped product: D[n][m] -- W fragment is the A operand, A fragment the B operand: the lane owns row m = 32 i + l31 of the wave tile
  auto mma = [&](int set) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[set][j]), __builtin_bit_cast(bf16x8, fa[set][i]), acc[i][j], 0, 0, 0);
  };

  if constexpr (BREG) {
    // ---------------- W through registers (see the header): persistent walk, one output tile after the other, nk even ----------------
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 fwg[2][4][4];                           // [k-tile parity][k-step][32-column block j]
    const unsigned vlane = (unsigned)lane * 16u;
    const size_t jstride = (size_t)nk * 4096;      // bytes between the packed panels of consecutive 32-column blocks
    const int stride_b = (MODE & 2) ? (int)gridDim.x : total;
    int vb = blockIdx.x;
    if (vb >= total) return;
    auto bases = [&](int v, const unsigned char*& ab, const unsigned char*& wpb, int& m0_, int& n0_) {
      const int logical = xcd_remap(v, total);
      m0_ = (logical / nt) * 256; n0_ = (logical % nt) * 256;
      ab = reinterpret_cast<const unsigned char*>(A) + (size_t)m0_ * lda * 2;
      wpb = reinterpret_cast<const unsigned char*>(Wp) + (size_t)((n0_ + wc * 128) / 32) * jstride;
    };
    const unsigned char *ab, *wpb, *abn, *wpbn;
    int m0b, n0b, m0bn, n0bn;
    bases(vb, ab, wpb, m0b, n0b);
    // prologue: A k-tile 0 -> stage 0 (8 pieces), W k-tile 0 -> register set 0 (16 loads)
#pragma unroll
    for (int i = 0; i < 8; ++i) W4_DMA1(voffA, ab + i * a_step, lds_wave + (unsigned)(i * 4096))
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j) W4_GLD(fwg[0][ks][j], vlane, wpb + j * jstride + ks * 1024)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto read_a = [&](int set, int stage, int ks) {
      const int co = ((2 * ks + hi) ^ sw) * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[set][i] = *reinterpret_cast<const uint4*>(a_rd + stage * STAGE + i * 4096 + co);
    };
    read_a(0, 0, 0);
    int par = 0;
    for (;;) {
      const int vn = vb + stride_b;
      const bool has_next = vn < total;
      if (has_next) bases(vn, abn, wpbn, m0bn, n0bn); else { abn = ab; wpbn = wpb; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      // one k-tile with compile-time register parity P: MFMAs on fa (LDS) x fwg[P]; fillers: the 4 A fragment reads of the next k-step,
      // the 16 W loads of the NEXT k-tile into fwg[P ^ 1] (6 + 5 + 5 in k-steps 0-2), its 8 A pieces by LDS-DMA (3 + 3 + 2)
      auto ktile = [&](auto pc, int kt) __attribute__((always_inline)) {
        constexpr int P = decltype(pc)::value;
        const int cur = par;
        const bool own = kt + 1 < nk;
        const unsigned char* a_n = own ? ab + (size_t)(kt + 1) * 128 : abn;
        const unsigned char* w_n = own ? wpb + (size_t)(kt + 1) * 4096 : wpbn;
        const unsigned lds_n = lds_wave + (unsigned)((cur ^ 1) * STAGE);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int set = ks & 1, nset = set ^ 1;
#pragma unroll
          for (int idx = 0; idx < 16; ++idx) {
            const int i = idx & 3, j = idx >> 2;
            if (ks == 3 && idx == 8) {
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next k-tile's A pieces AND its 16 W fragments have landed
              __builtin_amdgcn_s_barrier();
              W4_FENCE
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fwg[P][ks][j]), __builtin_bit_cast(bf16x8, fa[set][i]), acc[i][j], 0, 0, 0);
            W4_FENCE
            // ---- fillers
            if (ks < 3 ? idx < 4 : (idx >= 8 && idx < 12)) {          // A fragments of the next k-step (of the next k-tile's k-step 0 after the barrier)
              const int rd = ks < 3 ? idx : idx - 8;
              const int stage_r = ks < 3 ? cur : (cur ^ 1);
              const int ks_r = ks < 3 ? ks + 1 : 0;
              const int co = ((2 * ks_r + hi) ^ sw) * 16;
              fa[nset][rd] = *reinterpret_cast<const uint4*>(a_rd + stage_r * STAGE + rd * 4096 + co);
            }
            if (ks < 3) {
              const int g0 = ks == 0 ? 0 : ks == 1 ? 6 : 11, ng = ks == 0 ? 6 : 5;     // W loads of this k-step: 6 + 5 + 5 = 16
              if (idx >= 4 && idx - 4 < ng) {
                const int g = g0 + idx - 4;                       // 0..15 -> (k-step g >> 2, column block g & 3) of the next k-tile
                W4_GLD(fwg[P ^ 1][g >> 2][g & 3], vlane, w_n + (size_t)(g & 3) * jstride + (g >> 2) * 1024)
              }
              const int p0 = ks == 0 ? 0 : ks == 1 ? 3 : 6, np = ks == 2 ? 2 : 3;      // A pieces: 3 + 3 + 2 = 8
              if (idx >= 12 && idx - 12 < np) {
                const int pi = p0 + idx - 12;
                W4_DMA1(voffA, a_n + pi * a_step, lds_n + (unsigned)(pi * 4096))
              }
            }
            W4_FENCE
          }
        }
        par ^= 1;
      };
      for (int kt = 0; kt < nk; kt += 2) {
        ktile(std::integral_constant<int, 0>{}, kt);
        ktile(std::integral_constant<int, 1>{}, kt + 1);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc[i][j]));
      W4_FENCE
      const bool store = (MODE & 1) ? (acc[0][0][0] == 123456.789f) : true;
      if (store) {      // direct 8-byte stores (the epilogue is not what this mode measures)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          u16* crow = C + (size_t)(m0b + wr * 128 + 32 * i + l31) * ldc + n0b + wc * 128 + 4 * hi;
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 v4 = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
              *reinterpret_cast<uint2*>(crow + 32 * j + 8 * g) = __builtin_bit_cast(uint2, __builtin_convertvector(v4, bf16x4));
            }
        }
      }
      if (!has_next) break;
      vb = vn; ab = abn; wpb = wpbn; m0b = m0bn; n0b = n0bn;
    }
    return;
  }
  // The k-tiles of consecutive output tiles form ONE stream (as in the shipped persistent kernel): the last k-tile of an output tile
  // requests the FIRST k-tile of the block's next output tile (or, for the very last one, its own first k-tile again: valid memory,
  // never read), so there is one prologue per block and one copy of the k-loop body without a branch.  The epilogue uses no LDS.
  const int stride = (MODE & 2) ? (int)gridDim.x : total;    // persistent: a block walks v = blockIdx.x + i gridDim.x
  auto tile_bases = [&](int v, const unsigned char*& ab, const unsigned char*& wb, int& m0, int& n0) {
    const int logical = xcd_remap(v, total);
    m0 = (logical / nt) * 256; n0 = (logical % nt) * 256;
    ab = reinterpret_cast<const unsigned char*>(A) + (size_t)m0 * lda * 2;
    wb = reinterpret_cast<const unsigned char*>(W) + (size_t)n0 * ldw * 2;
  };
  int v = blockIdx.x;
  if (v >= total) return;
  const unsigned char *a_base, *w_base, *a_next, *w_next;
  int m0, n0, m0n, n0n;
  tile_bases(v, a_base, w_base, m0, n0);
  // prologue (once per block): k-tile 0 -> stage 0
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    W4_DMA1(voffA, a_base + i * a_step, lds_wave + (unsigned)(i * 4096))
    W4_DMA1(voffW, w_base + i * w_step, lds_wave + (unsigned)(WOFF + i * 4096))
  }
  if (MODE & 4) {                                // ablation without DMA: both stages hold real operands
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      W4_DMA1(voffA, a_base + i * a_step + 128, lds_wave + (unsigned)(STAGE + i * 4096))
      W4_DMA1(voffW, w_base + i * w_step + 128, lds_wave + (unsigned)(STAGE + WOFF + i * 4096))
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  read_frags(0, 0, 0);
  if (MODE & 8) read_frags(1, 0, 1);             // ablation without fragment reads: both register sets hold real operands
  int par = 0;                                   // stage that holds the current k-tile of the stream

  for (;;) {
    const int vn = v + stride;
    const bool has_next = vn < total;
    if (has_next) tile_bases(vn, a_next, w_next, m0n, n0n); else { a_next = a_base; w_next = w_base; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int kt = 0; kt < nk; ++kt) {
      const int cur = par;
      const bool own = kt + 1 < nk;
      const unsigned char* a_n = own ? a_base + (size_t)(kt + 1) * 128 : a_next;
      const unsigned char* w_n = own ? w_base + (size_t)(kt + 1) * 128 : w_next;
      const unsigned lds_n = lds_wave + (unsigned)((cur ^ 1) * STAGE);
      // One filler per MFMA gap (the guide: <= 5 single-issue instructions hide under a 32-cycle MFMA of a lone wave):
      //   k-steps 0-2: gaps 0-7 the eight fragment reads of the next k-step, gaps 8.. the LDS-DMA pieces of the next k-tile (6 + 5 + 5);
      //   k-step 3:    gaps 0-7 nothing, then vmcnt(0) + barrier (the next stage is published, this one is free), gaps 8-15 the fragment
      //                reads of k-step 0 of the next k-tile.
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int set = ks & 1, nset = set ^ 1;
        const int p0 = ks == 0 ? 0 : ks == 1 ? 6 : 11, np = ks == 0 ? 6 : 5;      // pieces of this k-step (ks < 3)
#pragma unroll
        for (int idx = 0; idx < 16; ++idx) {
          const int i = idx & 3, j = idx >> 2;
          if (ks == 3 && idx == 8) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            W4_FENCE
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[set][j]), __builtin_bit_cast(bf16x8, fa[set][i]), acc[i][j], 0, 0, 0);
          W4_FENCE
          // ---- the filler of this gap
          const int rd = ks < 3 ? idx : idx - 8;                 // fragment read 0..7: 0-3 = A rows i, 4-7 = W rows j
          if (!(MODE & 8) && (ks < 3 ? idx < 8 : idx >= 8)) {
            const int stage_r = ks < 3 ? cur : (cur ^ 1);
            const int ks_r = ks < 3 ? ks + 1 : 0;
            const int co = ((2 * ks_r + hi) ^ sw) * 16;
            if (rd < 4) fa[nset][rd] = *reinterpret_cast<const uint4*>(a_rd + stage_r * STAGE + rd * 4096 + co);
            else fw[nset][rd - 4] = *reinterpret_cast<const uint4*>(w_rd + stage_r * STAGE + (rd - 4) * 4096 + co);
          }
          if (!(MODE & 4) && ks < 3 && idx >= 8 && idx - 8 < np) {
            const int pc = p0 + idx - 8, pi = pc >> 1;
            if (pc & 1) { W4_DMA1(voffW, w_n + pi * w_step, lds_n + (unsigned)(WOFF + pi * 4096)) }
            else { W4_DMA1(voffA, a_n + pi * a_step, lds_n + (unsigned)(pi * 4096)) }
          }
          W4_FENCE
        }
      }
      par ^= 1;
    }
    // keep the accumulators in their AGPRs until the k-loop is over
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc[i][j]));
    W4_FENCE

    // ---- epilogue (no LDS): acc[i][j][r] = C[m0 + wr 128 + 32 i + l31][n0 + wc 128 + 32 j + 8 (r / 4) + 4 hi + (r % 4)]
    const bool store = (MODE & 1) ? (acc[0][0][0] == 123456.789f) : true;     // MODE bit 0: main loop only
    if (store && (MODE & 16)) {
      // LDS-transposed epilogue: the stage that was read last (par ^ 1 after the loop's flip = the one NOT holding the next k-tile) is
      // idle; every wave owns 8.5 KB of it: 32 rows x (128 columns + 8 pad) bf16.  Per 32-row slab: the lane (row l31) writes its
      // sixteen 4-column groups as ds_write_b64, then the wave reads whole 256-byte rows back (16 lanes x 16 B) and stores them.
      unsigned char* slab = smem + (par ^ 1) * STAGE + wave * 8704;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v4 = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            *reinterpret_cast<uint2*>(slab + l31 * 272 + (32 * j + 8 * g + 4 * hi) * 2) = __builtin_bit_cast(uint2, __builtin_convertvector(v4, bf16x4));
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        uint4 rowv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) rowv[t] = *reinterpret_cast<const uint4*>(slab + (4 * t + (lane >> 4)) * 272 + (lane & 15) * 16);
#pragma unroll
        for (int t = 0; t < 8; ++t)
          *reinterpret_cast<uint4*>(C + (size_t)(m0 + wr * 128 + 32 * i + 4 * t + (lane >> 4)) * ldc + n0 + wc * 128 + (lane & 15) * 8) = rowv[t];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
      __syncthreads();     // the next k-loop's DMA (any wave's pieces) overwrites this stage
    } else if (store) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u16* crow = C + (size_t)(m0 + wr * 128 + 32 * i + l31) * ldc + n0 + wc * 128 + 4 * hi;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v4 = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            *reinterpret_cast<uint2*>(crow + 32 * j + 8 * g) = __builtin_bit_cast(uint2, __builtin_convertvector(v4, bf16x4));
          }
      }
    }
    if (!has_next) break;
    v = vn; a_base = a_next; w_base = w_next; m0 = m0n; n0 = n0n;
  }
}

// pseudo-random bf16 values in [-scale/2, scale/2)
__global__ void init_kernel(u16* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ (unsigned)(i >> 32) * 40503u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    const float f = ((h >> 8) & 0xffff) / 65536.0f - 0.5f;
    p[i] = (u16)(__float_as_uint(f * scale) >> 16);       // truncation is fine for test data
  }
}

// W (N,K) row-major -> fragment order: [n / 32][k / 64][k-step 0..3][lane 0..63][8]: lane (l31, hi) of k-step ks holds W[32 nb + l31][64 kt + 8 (2 ks + hi) .. +7]
__global__ void pack_w_kernel(const u16* __restrict__ W, int ldw, int N, int K, u16* __restrict__ Wp) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 16-byte piece
  const size_t pieces = (size_t)N * K / 8;
  if (i >= pieces) return;
  const int lane = (int)(i & 63);
  const size_t f = i >> 6;                       // fragment index = (nb * nk + kt) * 4 + ks
  const int ks = (int)(f & 3);
  const int nk = K / 64;
  const int kt = (int)((f >> 2) % nk);
  const int nb = (int)((f >> 2) / nk);
  const int n = nb * 32 + (lane & 31), k = kt * 64 + 8 * (2 * ks + (lane >> 5));
  *reinterpret_cast<uint4*>(Wp + i * 8) = *reinterpret_cast<const uint4*>(W + (size_t)n * ldw + k);
}

// naive reference on a sample of rows
__global__ void ref_kernel(const u16* A, int lda, const u16* W, int ldw, float* out, const int* rows, int nrows, int N, int K) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int ri = blockIdx.y;
  if (n >= N || ri >= nrows) return;
  const int m = rows[ri];
  float s = 0.f;
  for (int k = 0; k < K; ++k) {
    const float a = __uint_as_float((unsigned)A[(size_t)m * lda + k] << 16), w = __uint_as_float((unsigned)W[(size_t)n * ldw + k] << 16);
    s += a * w;
  }
  out[(size_t)ri * N + n] = s;
}

static float bf2f(u16 h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

extern "C" double gemm_w4(int M, int N, int K, int mode, int iters, double* max_err) {
  if (M % 256 || N % 256 || K % 64 || K < 128) return -1.0;
  u16 *A, *W, *C;
  const size_t na = (size_t)M * K, nw = (size_t)N * K, nc = (size_t)M * N;
  if (hipMalloc((void**)&A, na * 2) != hipSuccess || hipMalloc((void**)&W, nw * 2) != hipSuccess || hipMalloc((void**)&C, nc * 2) != hipSuccess) return -2.0;
  const float zs = (mode & 32) ? 0.f : 1.f;      // mode bit 5: all-zero operands (the power argument: same instructions, no toggling)
  hipLaunchKernelGGL(init_kernel, dim3(4096), dim3(256), 0, 0, A, na, 12345u, 2.0f * zs);
  hipLaunchKernelGGL(init_kernel, dim3(4096), dim3(256), 0, 0, W, nw, 777u, 0.1f * zs);
  (void)hipMemset(C, 0, nc * 2);
  u16* Wp = nullptr;
  if (mode & 64) {
    if ((K / 64) % 2) return -1.0;
    if (hipMalloc((void**)&Wp, nw * 2) != hipSuccess) return -2.0;
    hipLaunchKernelGGL(pack_w_kernel, dim3((unsigned)((nw / 8 + 255) / 256)), dim3(256), 0, 0, W, K, N, K, Wp);
  }
  constexpr int LDS = 2 * 65536;
  int ncu = 256;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  ncu = (ncu / 8) * 8;
  const int total = (M / 256) * (N / 256);
  const int grid = (mode & 2) ? (total < ncu ? total : ncu) : total;
#define W4_CASE(MD)                                                                                                          \
  case MD:                                                                                                                   \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w4_kernel<MD>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
    hipLaunchKernelGGL(gemm_w4_kernel<MD>, dim3(grid), dim3(256), LDS, 0, A, K, W, K, C, N, M, N, K, Wp);                    \
    break;
  auto launch = [&]() {
    switch (mode & (31 | 64)) {
      W4_CASE(0) W4_CASE(1) W4_CASE(2) W4_CASE(3) W4_CASE(7) W4_CASE(11) W4_CASE(15) W4_CASE(16) W4_CASE(18) W4_CASE(66) W4_CASE(67)
      default: break;
    }
  };
  launch();
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "gemm_w4: launch failed: %s\n", hipGetErrorString(hipGetLastError())); return -3.0; }
  if (max_err) {
    *max_err = -1.0;
    if (!(mode & 1)) {
      const int nrows = 24;
      std::vector<int> rows(nrows);
      for (int i = 0; i < nrows; ++i) rows[i] = (int)(((long)i * 2654435761u) % (unsigned)M);
      rows[0] = 0; rows[1] = M - 1; rows[2] = 255; rows[3] = 256;
      int* drows; float* dref;
      (void)hipMalloc((void**)&drows, nrows * 4); (void)hipMalloc((void**)&dref, (size_t)nrows * N * 4);
      (void)hipMemcpy(drows, rows.data(), nrows * 4, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(ref_kernel, dim3((N + 255) / 256, nrows), dim3(256), 0, 0, A, K, W, K, dref, drows, nrows, N, K);
      std::vector<float> href((size_t)nrows * N);
      std::vector<u16> hc(N);
      (void)hipMemcpy(href.data(), dref, href.size() * 4, hipMemcpyDeviceToHost);
      double worst = 0.0;
      for (int i = 0; i < nrows; ++i) {
        (void)hipMemcpy(hc.data(), C + (size_t)rows[i] * N, (size_t)N * 2, hipMemcpyDeviceToHost);
        for (int n = 0; n < N; ++n) {
          const double ref = href[(size_t)i * N + n], got = bf2f(hc[n]);
          const double e = fabs(got - ref) / (fabs(ref) + 0.05);
          if (e > worst) worst = e;
        }
      }
      *max_err = worst;
      (void)hipFree(drows); (void)hipFree(dref);
    }
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  for (int it = 0; it < iters; ++it) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(A); (void)hipFree(W); (void)hipFree(C); if (Wp) (void)hipFree(Wp);
  return 2.0 * M * N * K * iters / (ms * 1e-3) / 1e12;
}
