// Instruction-mix microbenchmark for the 16-bit attention inner loop on MI355X (r02): which arrangement of the SAME work -- per
// 32-query x 64-key tile 16 v_mfma_f32_32x32x16_bf16, 32 v_exp_f32, 16 v_cvt_pk_bf16_f32, the row-sum adds, 16 ds_read_b128 --
// keeps the matrix pipe busy?  No global memory inside the loop (K / V^T fragments are re-read from a static LDS tile), so this is
// the ceiling of the loop STRUCTURE, to be compared with the real kernel (attn_h16.hip) and its PMC counters.
//
//   QB    query blocks of 32 per wave (1 = the shipped kernel's wave tile, 2 = 64-query wave tile: fragment reads shared)
//   MODE  0 = phase-serial (S = K Q^T for the whole tile | softmax | O += V P), the shipped structure
//         1 = software-pipelined, one MFMA per slot followed by ITS share of the softmax of the previous tile
//             [mfma, exp, exp, cvt_pk, add, add] pinned with sched_barrier(0); P*V runs one quarter behind the exponentials
//         2 = as 1 with the row sums on v_pk_add_f32 (half the add instructions; the guide prices packed fp32 beside MFMAs as a loss)
//         3 = as 1 without the pins (compiler's order)
//         4 = as 1 with a block barrier per tile;  6 = as 3 with a block barrier per tile;  7 = the asm slot blocks of 1 without the pins
//         8 = as 3 but P*V uses the P of the SAME quarter (no lag);  9 / 10 = as 3 with one block barrier per 4 / 2 tiles;  11 = 8 + barrier per 4 tiles
//         13 = 12 with the loads only, 14 = 12 with the ds_writes only, 15 = 12 with every block streaming the SAME 2 MB (L2 hits)
//         12 = as 10 + the register-staged K / V^T stream of the real kernel (4 global_load_dwordx4 at the top of a tile pair, 4 ds_write_b128 at its end)
//   BAR   (mode 0) one __syncthreads() per tile, as the shipped kernel has
// extern "C" double attn_mix(int qb, int mode, int waves, int occ, int lds_pad_bytes, int blocks, int tiles, double* ms)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
#define HLD 72

__device__ __forceinline__ uint32_t cvt_pk(float a, float b) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float ex2(float a) { return __builtin_amdgcn_exp2f(a); }
__device__ __forceinline__ float fadd(float a, float b) {   // a single v_add_f32 the SLP vectoriser cannot pack (empty asm: no code,
  float r = a + b;                                           // and the compiler still sees the add for its v_exp_f32 -> VALU hazard)
  asm("" : "+v"(r));
  return r;
}
__device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
#define SB __builtin_amdgcn_sched_barrier(0);

template <int QB, int MODE, int NW, int OCC>
__global__ __launch_bounds__(NW * 64, OCC) void mix_kernel(const u16* __restrict__ src, float* __restrict__ out, int tiles,
                                                            const u16* __restrict__ stream) {
  extern __shared__ __attribute__((aligned(16))) u16 smem[];   // [K 64x72][V 64x72] (+ a 4-tile parking area in mode 12) + padding that sets blocks per CU
  u16* Ks = smem;
  u16* Vs = smem + 64 * HLD;
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  for (int i = tid; i < 2 * 64 * HLD; i += blockDim.x) smem[i] = src[(blockIdx.x * 131 + i) & 0xffff];
  __syncthreads();
  const int lrow = l31 * HLD + 8 * hi;
  uint4 qf[QB][4];
#pragma unroll
  for (int b = 0; b < QB; ++b)
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[b][s] = *reinterpret_cast<const uint4*>(src + ((tid * 64 + b * 16384 + 16 * s) & 0xfff8));
  f32x16 o0[QB], o1[QB], zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
#pragma unroll
  for (int b = 0; b < QB; ++b) { o0[b] = zero16; o1[b] = zero16; }
  float ls[QB][4];
#pragma unroll
  for (int b = 0; b < QB; ++b) { ls[b][0] = 0.f; ls[b][1] = 0.f; ls[b][2] = 0.f; ls[b][3] = 0.f; }

  uint32_t sink = 0;
  if (MODE == 0 || MODE == 5) {
    for (int t = 0; t < tiles; ++t) {
      f32x16 s0[QB], s1[QB];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const uint4 k0 = *reinterpret_cast<const uint4*>(Ks + lrow + 16 * s);
        const uint4 k1 = *reinterpret_cast<const uint4*>(Ks + lrow + 32 * HLD + 16 * s);
#pragma unroll
        for (int b = 0; b < QB; ++b) {
          s0[b] = mfma(k0, qf[b][s], s == 0 ? zero16 : s0[b]);
          s1[b] = mfma(k1, qf[b][s], s == 0 ? zero16 : s1[b]);
        }
      }
#pragma unroll
      for (int b = 0; b < QB; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s0[b][r] = ex2(s0[b][r]);
          s1[b][r] = ex2(s1[b][r]);
          ls[b][r & 3] = fadd(ls[b][r & 3], s0[b][r]);
          ls[b][(r + 2) & 3] = fadd(ls[b][(r + 2) & 3], s1[b][r]);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int rb = 8 * (ks & 1);
        const uint4 v0 = *reinterpret_cast<const uint4*>(Vs + lrow + 16 * ks);
        const uint4 v1 = *reinterpret_cast<const uint4*>(Vs + lrow + 32 * HLD + 16 * ks);
#pragma unroll
        for (int b = 0; b < QB; ++b) {
          const f32x16& sc = (ks >> 1) ? s1[b] : s0[b];
          uint4 pb;
          pb.x = cvt_pk(sc[rb + 0], sc[rb + 1]); pb.y = cvt_pk(sc[rb + 2], sc[rb + 3]);
          pb.z = cvt_pk(sc[rb + 4], sc[rb + 5]); pb.w = cvt_pk(sc[rb + 6], sc[rb + 7]);
          o0[b] = mfma(v0, pb, o0[b]);
          o1[b] = mfma(v1, pb, o1[b]);
        }
      }
      if (MODE == 0) __syncthreads();
    }
  } else {
    // software-pipelined: sc = finished scores of tile t (being exponentiated), sn = scores of tile t+1 (being accumulated)
    f32x16 sa0[QB], sa1[QB], sb0[QB], sb1[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) { sa0[b] = zero16; sa1[b] = zero16; sb0[b] = zero16; sb1[b] = zero16; }
    uint4 pb[QB];                                  // P of the previous quarter (the P*V MFMAs run one quarter behind)
#pragma unroll
    for (int b = 0; b < QB; ++b) pb[b] = make_uint4(0, 0, 0, 0);
#define PIN if (MODE == 1 || MODE == 2 || MODE == 4) { SB }
    // one slot: an MFMA was just issued; now 2 exponentials, 1 conversion, 2 adds of quarter (SC, RB), elements E..E+1
#define SOFT2(SC, RB, E, PW, L0, L1)                                                     \
  {                                                                                      \
    if (MODE == 2) {                                                                     \
      const float e0_ = ex2(SC[(RB) + (E)]), e1_ = ex2(SC[(RB) + (E) + 1]);              \
      PW = cvt_pk(e0_, e1_);                                                             \
      f32x2 p_ = {L0, L1}; p_ += f32x2{e0_, e1_}; L0 = p_.x; L1 = p_.y;                  \
    } else if (MODE == 3 || MODE == 6 || MODE >= 8) {  /* incl. 12 */                                                 \
      const float e0_ = ex2(SC[(RB) + (E)]), e1_ = ex2(SC[(RB) + (E) + 1]);              \
      PW = cvt_pk(e0_, e1_);                                                             \
      L0 = fadd(L0, e0_); L1 = fadd(L1, e1_);                                            \
    } else {                                                                             \
      float e0_, e1_;     /* one opaque block: order fixed, 1 instruction between every v_exp_f32 and its first use (trans hazard) */ \
      asm volatile("v_exp_f32 %0, %5\n\tv_exp_f32 %1, %6\n\tv_add_f32 %2, %2, %0\n\tv_add_f32 %3, %3, %1\n\tv_cvt_pk_bf16_f32 %4, %0, %1" \
                   : "=&v"(e0_), "=&v"(e1_), "+v"(L0), "+v"(L1), "=v"(PW) : "v"(SC[(RB) + (E)]), "v"(SC[(RB) + (E) + 1]));          \
    }                                                                                    \
  }
#define TILE(SC0, SC1, SN0, SN1)                                                         \
  {                                                                                      \
    uint4 fk0_ = *reinterpret_cast<const uint4*>(Ks + lrow), fk1_ = *reinterpret_cast<const uint4*>(Ks + lrow + 32 * HLD); \
    uint4 fv0_ = *reinterpret_cast<const uint4*>(Vs + lrow), fv1_ = *reinterpret_cast<const uint4*>(Vs + lrow + 32 * HLD); \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                   \
      uint4 nk0_, nk1_, nv0_, nv1_;                                                      \
      {                                                                                  \
        const int o_ = 16 * ((ks + 1) & 3);                                              \
        nk0_ = *reinterpret_cast<const uint4*>(Ks + lrow + o_); nk1_ = *reinterpret_cast<const uint4*>(Ks + lrow + 32 * HLD + o_); \
        nv0_ = *reinterpret_cast<const uint4*>(Vs + lrow + o_); nv1_ = *reinterpret_cast<const uint4*>(Vs + lrow + 32 * HLD + o_); \
      }                                                                                  \
      const int rb_ = 8 * (ks & 1);                                                      \
      uint4 np_[QB];                                                                     \
      _Pragma("unroll") for (int b = 0; b < QB; ++b) {                                   \
        PIN                                                                              \
        SN0[b] = mfma(fk0_, qf[b][ks], ks == 0 ? zero16 : SN0[b]);                       \
        if ((ks >> 1) == 0) SOFT2(SC0[b], rb_, 0, np_[b].x, ls[b][0], ls[b][1]) else SOFT2(SC1[b], rb_, 0, np_[b].x, ls[b][0], ls[b][1]) \
        PIN                                                                              \
        SN1[b] = mfma(fk1_, qf[b][ks], ks == 0 ? zero16 : SN1[b]);                       \
        if ((ks >> 1) == 0) SOFT2(SC0[b], rb_, 2, np_[b].y, ls[b][2], ls[b][3]) else SOFT2(SC1[b], rb_, 2, np_[b].y, ls[b][2], ls[b][3]) \
        PIN                                                                              \
        if (MODE != 8 && MODE != 11) o0[b] = mfma(fv0_, pb[b], o0[b]);                   \
        if ((ks >> 1) == 0) SOFT2(SC0[b], rb_, 4, np_[b].z, ls[b][0], ls[b][1]) else SOFT2(SC1[b], rb_, 4, np_[b].z, ls[b][0], ls[b][1]) \
        PIN                                                                              \
        if (MODE != 8 && MODE != 11) o1[b] = mfma(fv1_, pb[b], o1[b]);                   \
        if ((ks >> 1) == 0) SOFT2(SC0[b], rb_, 6, np_[b].w, ls[b][2], ls[b][3]) else SOFT2(SC1[b], rb_, 6, np_[b].w, ls[b][2], ls[b][3]) \
        if (MODE == 8 || MODE == 11) { o0[b] = mfma(fv0_, np_[b], o0[b]); o1[b] = mfma(fv1_, np_[b], o1[b]); } \
      }                                                                                  \
      PIN                                                                                \
      _Pragma("unroll") for (int b = 0; b < QB; ++b) pb[b] = np_[b];                     \
      fk0_ = nk0_; fk1_ = nk1_; fv0_ = nv0_; fv1_ = nv1_;                                \
    }                                                                                    \
    if (MODE == 4 || MODE == 6) __syncthreads();                                                      \
  }
    // mode 12: the K / V^T stream of the real kernel: 2 MB per (sample, head) shared by 32 blocks, 16 KB per tile, register-staged
    const int srow = tid >> 3, sch = (tid & 7) * 8;
    const u16* sp = stream + (MODE == 15 ? (size_t)0 : (size_t)((blockIdx.x >> 5) & 7) * (1 << 20)) + srow * 64 + sch;
    u16* park = smem + 2 * 64 * HLD + srow * HLD + sch;
    uint4 g0 = make_uint4(1, 2, 3, 4), g1 = g0, g2 = g0, g3 = g0;
    for (int t = 0; t < tiles; t += 2) {
      if (MODE == 12 || MODE == 13 || MODE == 15) {
        const u16* q = sp + (size_t)((t * 2) & 127) * 4096;
        g0 = *reinterpret_cast<const uint4*>(q); g1 = *reinterpret_cast<const uint4*>(q + 4096);
        g2 = *reinterpret_cast<const uint4*>(q + 8192); g3 = *reinterpret_cast<const uint4*>(q + 12288);
        SB
      }
      TILE(sa0, sa1, sb0, sb1)
      TILE(sb0, sb1, sa0, sa1)
      if (MODE == 13) { SB sink ^= g0.x ^ g1.y ^ g2.z ^ g3.w; }
      if (MODE == 12 || MODE == 14 || MODE == 15) {
        SB
        *reinterpret_cast<uint4*>(park) = g0; *reinterpret_cast<uint4*>(park + 64 * HLD) = g1;
        *reinterpret_cast<uint4*>(park + 128 * HLD) = g2; *reinterpret_cast<uint4*>(park + 192 * HLD) = g3;
      }
      if (MODE == 10 || MODE >= 12 || ((MODE == 9 || MODE == 11) && (t & 2))) __syncthreads();
    }
  }
  float acc = (float)(sink & 1u);
#pragma unroll
  for (int b = 0; b < QB; ++b) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += o0[b][r] + o1[b][r];
    acc += ls[b][0] + ls[b][1] + ls[b][2] + ls[b][3];
  }
  out[(size_t)blockIdx.x * blockDim.x + tid] = acc;
}

template <int QB, int MODE, int NW, int OCC>
static double run(int lds_pad, int blocks, int tiles, double* ms_out) {
  const int waves = NW;
  u16* src; float* out; u16* stream;
  (void)hipMalloc(&stream, (size_t)8 << 21);
  (void)hipMemset(stream, 0x3c, (size_t)8 << 21);
  (void)hipMalloc(&src, 65536 * 2 + 4096);
  (void)hipMalloc(&out, (size_t)blocks * waves * 64 * 4);
  {
    u16* h = new u16[65536 + 2048];
    uint32_t s = 12345u;
    for (int i = 0; i < 65536 + 2048; ++i) { s = s * 1664525u + 1013904223u; h[i] = (u16)(0x3c00u + ((s >> 16) & 0x3ffu) + ((s >> 31) << 15)); }
    (void)hipMemcpy(src, h, (65536 + 2048) * 2, hipMemcpyHostToDevice);
    delete[] h;
  }
  const size_t lds = 2 * 64 * HLD * 2 + (MODE >= 12 ? 4 * 64 * HLD * 2 : 0) + (size_t)lds_pad;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mix_kernel<QB, MODE, NW, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((mix_kernel<QB, MODE, NW, OCC>), dim3(blocks), dim3(waves * 64), lds, 0, src, out, tiles, stream);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((mix_kernel<QB, MODE, NW, OCC>), dim3(blocks), dim3(waves * 64), lds, 0, src, out, tiles, stream);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 3.f;
  (void)hipFree(src); (void)hipFree(out); (void)hipFree(stream);
  if (ms_out) *ms_out = ms;
  const double flops = (double)blocks * waves * tiles * (16.0 * QB) * 2.0 * 32 * 32 * 16;
  return flops / (ms * 1e-3) * 1e-12;
}

// (qb, mode, waves per block, waves per SIMD the register budget allows); blocks per CU = min(register limit, LDS limit via lds_pad)
extern "C" double attn_mix(int qb, int mode, int waves, int occ, int lds_pad, int blocks, int tiles, double* ms) {
#define CASE(Q, M, W, O) if (qb == Q && mode == M && waves == W && occ == O) return run<Q, M, W, O>(lds_pad, blocks, tiles, ms);
  CASE(1, 0, 8, 4) CASE(1, 5, 8, 4) CASE(1, 0, 4, 4)
  CASE(1, 1, 8, 2) CASE(1, 2, 8, 2) CASE(1, 3, 8, 2) CASE(1, 4, 8, 2)
  CASE(1, 1, 4, 3) CASE(1, 3, 4, 3) CASE(1, 1, 4, 2) CASE(1, 1, 4, 1)
  CASE(2, 0, 4, 2) CASE(2, 5, 4, 2) CASE(2, 0, 8, 2)
  CASE(2, 1, 4, 1) CASE(2, 2, 4, 1) CASE(2, 3, 4, 1) CASE(2, 4, 4, 1)
  CASE(2, 1, 4, 2) CASE(2, 3, 4, 2) CASE(2, 1, 8, 2)
  CASE(1, 12, 8, 2) CASE(1, 12, 4, 3) CASE(1, 13, 8, 2) CASE(1, 14, 8, 2) CASE(1, 15, 8, 2)
  CASE(1, 8, 8, 2) CASE(1, 9, 8, 2) CASE(1, 10, 8, 2) CASE(1, 11, 8, 2) CASE(1, 8, 4, 3) CASE(1, 9, 4, 3) CASE(1, 9, 4, 2) CASE(1, 11, 4, 3)
  CASE(1, 6, 8, 2) CASE(1, 7, 8, 2) CASE(1, 6, 4, 3) CASE(1, 7, 4, 3) CASE(1, 3, 4, 2) CASE(1, 3, 4, 1) CASE(2, 6, 4, 2) CASE(2, 7, 4, 2)
  return -1.0;
}
