// Few-token calls (round 6): the attention out-projection, its bias + residual add and the LayerNorm that follows it in ONE kernel
// (flow_model/layer.py:152-163: x = x + to_out(attn(norm(x))), then the next norm -- adaLN of the per-sample branch after the per-part
// branch, the FFN's affine LayerNorm after the per-sample branch).
//
// Why: a demo pair (2 x 1024 points) runs the out-projection as a 128 x 128-tile GEMM split over K into fp32 partial planes (10.3 us) and
// a combine + LayerNorm pass over those planes (5.9 us) -- both at their launch floors, 240 times per call each.  A LayerNorm needs whole
// rows, so the fused form gives every block 32 token rows x ALL 512 output columns: the block streams the whole 512 x 512 weight
// (512 KB, L2-resident) once.  A ring of LDS stages cannot keep that stream busy (96 KB in flight against ~1.2 us of L2 latency is
// ~80 GB/s per CU), registers can: every wave owns 64 output columns and loads ITS 64 weight rows straight into MFMA fragments
// (global_load_dwordx4, three k-tiles = 24 loads per lane in flight, ordinary loads the compiler counts by itself), the 32 x 512 activation tile
// sits in LDS (one LDS-DMA burst, swizzled as in gemm_h16.hip) -- no barrier inside the k-loop.  The product is SWAPPED
// (C^T = W A^T on v_mfma_f32_32x32x16): a lane owns one token column and 32 of a wave's 64 output columns, so bias, residual, the stored
// stream value and both LayerNorm sums are lane-local; the eight waves exchange two floats per token through LDS.
//   arithmetic per element = the unfused sequence's: (sum_k + bias) + residual, ONE (saturating) rounding when the stream is fp16, LayerNorm
//   of the STORED value with two-pass fp32 statistics, modulation in fp32, one rounding to the operand type.  The k-sum is one chain instead
//   of split-K planes and the statistics are summed in another order: same function, not bit-identical (tests: against the unfused
//   sequence at rounding level, against the oracle in the 16-bit deviation class).
#include <type_traits>
#include "half.h"
#include "kernels.h"

struct OutprojLnParams {
  const u16* A; int lda;            // attention output (M, 512) 16-bit
  const u16* W;                     // (512, 512) 16-bit in FRAGMENT order (launch_outproj_pack_h16)
  const float* bias;                // (512)
  void* h;                          // residual stream (M, 512), fp32 or fp16 (XH): read, rewritten in place
  u16* out;                         // LayerNorm output (M, 512) 16-bit
  int M;
  const float* gain; const float* shift; long row_stride; const int32_t* token_row; int add_one;      // as layernorm_h16_kernel
};

#define OPL_D 512
#define OPL_PD 4          // k-tiles of weight fragments in flight per wave (32 loads of 1 KB; 3: 14.6 us per launch, r06 call 22)

template <int DT, bool XH>
__global__ __launch_bounds__(512, 2) void outproj_ln_h16_kernel(OutprojLnParams p) {
  typedef typename H16<DT>::T8 T8;
  constexpr int NK = OPL_D / 64;                                        // 8 k-tiles
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NK * 4096 + 2 * 8 * 32 * 4];      // A tile [8 k-tiles][32 rows][128 B] + the two stat exchanges
  float* red = reinterpret_cast<float*>(smem + NK * 4096);             // [2][8 waves][32 tokens]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int m0 = blockIdx.x * 32;
  int tok = m0 + l31;
  const bool live = tok < p.M;
  tok = live ? tok : p.M - 1;

  // ---- the block's activation tile: wave w fetches k-tile w (four 1 KB pieces: rows 8 sub .. 8 sub + 7, slot ^ ((row >> 1) & 7))
  {
    const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + (unsigned)wave * 4096u);
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      const int row = 8 * sub + (lane >> 3);
      const int lslot = (lane & 7) ^ ((row >> 1) & 7);
      int r = m0 + row;
      r = r < p.M ? r : p.M - 1;
      const u16* src = p.A + (size_t)r * p.lda + 64 * wave + 8 * lslot;
      unsigned keep_;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep_) : "v"(src), "s"(lds_w + (unsigned)sub * 1024u) : "memory");
    }
  }
  // ---- this wave's weight rows: fragment (j, g) of k-tile kt = W[64 wave + 32 j + l31][64 kt + 16 g + 8 hi .. +7], stored in that order
  // (n-tile T = 2 wave + j, k-step S = 4 kt + g: 64 lanes x 16 B contiguous).  Read from the nn.Linear layout the same fragment is 32 rows x
  // 32 bytes: every load touches 32 cache lines for 1 KB, the lines are evicted from the 32 KB L1 between the four k-steps that share them
  // (192 KB of fragments are in flight per CU), and the kernel measured 23.7 us -- 26 GB/s per CU (r06 call 21).
  const u16* wfrag = p.W + ((size_t)(2 * wave) * 32 * 64 + lane) * 8;
  uint4 wq[OPL_PD][2][4];
  auto load_w = [&](uint4 (&dst)[2][4], int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        dst[j][g] = *reinterpret_cast<const uint4*>(wfrag + ((size_t)j * 32 + 4 * kt + g) * (64 * 8));
  };
#pragma unroll
  for (int s = 0; s < OPL_PD; ++s) load_w(wq[s], s);
  // the activation pieces were requested BEFORE the 8 OPL_PD weight loads above and loads retire in order: they have landed once at most
  // that many are outstanding (the compiler does not know about the LDS-DMA; its own counts for the weight registers stay correct)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * OPL_PD) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // the modulation row of this lane's token, requested here so that the gain / shift loads of the epilogue do not wait for it (an
  // UNCONDITIONAL load -- the launcher passes a readable dummy table when there is none and row_stride = 0: a conditional one makes the
  // compiler wait for it on the spot)
  int mrow32 = p.token_row[tok];

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int sw = (l31 >> 1) & 7;
#pragma unroll
  for (int kt = 0; kt < NK; ++kt) {
    uint4 af[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) af[g] = *reinterpret_cast<const uint4*>(smem + kt * 4096 + l31 * 128 + ((2 * g + hi) ^ sw) * 16);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[j] = H16<DT>::mfma(__builtin_bit_cast(T8, wq[kt % OPL_PD][j][g]), __builtin_bit_cast(T8, af[g]), acc[j]);
    // (without the fences the machine scheduler sinks these refills below the NEXT tiles' MFMAs and the stream collapses to two or three
    // loads in flight: the registers of tile kt are free as soon as its MFMAs have issued, and that is when tile kt + OPL_PD is requested)
    __builtin_amdgcn_sched_barrier(0);
    if (kt + OPL_PD < NK) load_w(wq[kt % OPL_PD], kt + OPL_PD);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue.  acc[j][r] = column n = 64 wave + 32 j + crow(r, hi) of token m0 + l31: groups of four consecutive columns.
  // Everything the epilogue reads is requested at once, in front of the two barriers of the statistics (one memory latency, not three)
  asm volatile("" : "+v"(mrow32));      // (first use HERE: hipcc otherwise sign-extends the value right behind the load and drains the weight stream for it)
  const long mrow = (long)mrow32;
  const float* gp = p.gain + mrow * p.row_stride;
  const float* bp = p.shift + mrow * p.row_stride;
  float4 bia[2][4], gg[2][4], bb[2][4];
  typename std::conditional<XH, uint2, float4>::type rr[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = 64 * wave + 32 * j + 8 * q + 4 * hi;
      if constexpr (XH) rr[j][q] = *reinterpret_cast<const uint2*>(reinterpret_cast<const u16*>(p.h) + (size_t)tok * OPL_D + n);
      else rr[j][q] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.h) + (size_t)tok * OPL_D + n);
      bia[j][q] = *reinterpret_cast<const float4*>(p.bias + n);
      gg[j][q] = *reinterpret_cast<const float4*>(gp + n);
      bb[j][q] = *reinterpret_cast<const float4*>(bp + n);
    }
  float x[2][16];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = 64 * wave + 32 * j + 8 * q + 4 * hi;
      const float4 b4 = bia[j][q];
      float v0 = acc[j][4 * q + 0] + b4.x, v1 = acc[j][4 * q + 1] + b4.y, v2 = acc[j][4 * q + 2] + b4.z, v3 = acc[j][4 * q + 3] + b4.w;
      if constexpr (XH) {
        u16* hp = reinterpret_cast<u16*>(p.h) + (size_t)tok * OPL_D + n;
        const typename H16<RAP_DT_F16>::T4 r4 = __builtin_bit_cast(typename H16<RAP_DT_F16>::T4, rr[j][q]);
        v0 += (float)r4[0]; v1 += (float)r4[1]; v2 += (float)r4[2]; v3 += (float)r4[3];
        const uint2 st = h16_pack4<RAP_DT_F16>(f16_sat(v0), f16_sat(v1), f16_sat(v2), f16_sat(v3));      // ONE saturating rounding: the stream value
        if (live) *reinterpret_cast<uint2*>(hp) = st;
        const typename H16<RAP_DT_F16>::T4 s4 = __builtin_bit_cast(typename H16<RAP_DT_F16>::T4, st);
        v0 = (float)s4[0]; v1 = (float)s4[1]; v2 = (float)s4[2]; v3 = (float)s4[3];                      // the LayerNorm sees the STORED value
      } else {
        float* hp = reinterpret_cast<float*>(p.h) + (size_t)tok * OPL_D + n;
        const float4 r4 = rr[j][q];
        v0 += r4.x; v1 += r4.y; v2 += r4.z; v3 += r4.w;
        if (live) *reinterpret_cast<float4*>(hp) = float4{v0, v1, v2, v3};
      }
      x[j][4 * q + 0] = v0; x[j][4 * q + 1] = v1; x[j][4 * q + 2] = v2; x[j][4 * q + 3] = v3;
      s += (v0 + v1) + (v2 + v3);
    }
  // two-pass statistics over the token's 512 columns: lane -> half-wave pair -> the eight waves (fixed order: deterministic)
  auto xhalf_sum = [](float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  };
  auto block_sum = [&](float v, float* slot) {
    v = xhalf_sum(v);
    if (hi == 0) slot[wave * 32 + l31] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) t += slot[w8 * 32 + l31];
    return t;
  };
  const float mean = block_sum(s, red) / (float)OPL_D;
  float qq = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float dlt = x[j][r] - mean; qq += dlt * dlt; }
  const float var = block_sum(qq, red + 8 * 32) / (float)OPL_D;
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  const float one = p.add_one ? 1.0f : 0.0f;
  if (!live) return;
  u16* orow = p.out + (size_t)tok * OPL_D;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = 64 * wave + 32 * j + 8 * q + 4 * hi;
      const float o0 = (x[j][4 * q + 0] - mean) * rstd * (one + gg[j][q].x) + bb[j][q].x;
      const float o1 = (x[j][4 * q + 1] - mean) * rstd * (one + gg[j][q].y) + bb[j][q].y;
      const float o2 = (x[j][4 * q + 2] - mean) * rstd * (one + gg[j][q].z) + bb[j][q].z;
      const float o3 = (x[j][4 * q + 3] - mean) * rstd * (one + gg[j][q].w) + bb[j][q].w;
      *reinterpret_cast<uint2*>(orow + n) = h16_pack4<DT>(o0, o1, o2, o3);
    }
}

// one thread per 16-byte fragment piece
__global__ __launch_bounds__(256) void outproj_pack_h16_kernel(const u16* __restrict__ W, int ldw, u16* __restrict__ packed) {
  const int id = blockIdx.x * 256 + threadIdx.x;          // ((T * 32 + S) * 64 + lane)
  if (id >= 16 * 32 * 64) return;
  const int lane = id & 63, S = (id >> 6) & 31, T = id >> 11;
  const int hi = lane >> 5, l31 = lane & 31;
  *reinterpret_cast<uint4*>(packed + (size_t)id * 8) = *reinterpret_cast<const uint4*>(W + (size_t)(32 * T + l31) * ldw + 16 * S + 8 * hi);
}
int launch_outproj_pack_h16(hipStream_t stream, const u16* W, int ldw, u16* packed) {
  if (!W || !packed || ldw < OPL_D || (ldw & 7)) return RAP_ERR_INVALID;
  hipLaunchKernelGGL(outproj_pack_h16_kernel, dim3(16 * 32 * 64 / 256), dim3(256), 0, stream, W, ldw, packed);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

rap_tuning_t g_rap_outproj_ln = 1;      // tuning key 21: 1 (default) = the rule below, 0 = never, 2 = every call of at most 8 192 token rows (A/B)

// does a few-token call of `rows` token rows (align_up(TP, 256)) take the fused out-projection + LayerNorm?  One block per 32 rows, every block
// streams the whole weight: it pays while the blocks fit the CUs with room to spare
bool outproj_ln_wanted(int dtype, long rows, int d, int K) {
  const int mode = g_rap_outproj_ln;
  if (!mode || (dtype != RAP_DT_BF16 && dtype != RAP_DT_F16) || d != OPL_D || K != OPL_D || rows <= 0) return false;
  return rows <= (mode == 2 ? 8192 : 4096);
}

int launch_outproj_ln_h16(hipStream_t stream, int dtype, const u16* A, int lda, const u16* W, const float* bias, void* h, int h_f16,
                          u16* out, int rows, int d, int K, const float* mod, long mod_stride, const int32_t* token_row, const float* gain,
                          const float* shift) {
  if (!A || !W || !bias || !h || !out || d != OPL_D || K != OPL_D || (lda & 7) || lda < K) return RAP_ERR_INVALID;
  if (!mod && (!gain || !shift)) return RAP_ERR_INVALID;
  if (rows <= 0) return RAP_OK;
  OutprojLnParams p{};
  p.A = A; p.lda = lda; p.W = W; p.bias = bias; p.h = h; p.out = out; p.M = rows;
  if (mod) { p.gain = mod; p.shift = mod + d; p.row_stride = mod_stride; p.token_row = token_row; p.add_one = 1; }
  else { p.gain = gain; p.shift = shift; p.row_stride = 0; p.token_row = nullptr; p.add_one = 0; }
  // (no table: any readable int32 per token will do, its value is multiplied by row_stride = 0 -- A holds 256 of them per row)
  if (!p.token_row) { p.token_row = reinterpret_cast<const int32_t*>(A); p.row_stride = 0; }
  const dim3 grid((unsigned)((rows + 31) / 32)), block(512);
  if (dtype == RAP_DT_BF16) {
    if (h_f16) hipLaunchKernelGGL((outproj_ln_h16_kernel<RAP_DT_BF16, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((outproj_ln_h16_kernel<RAP_DT_BF16, false>), grid, block, 0, stream, p);
  } else if (dtype == RAP_DT_F16) {
    if (h_f16) hipLaunchKernelGGL((outproj_ln_h16_kernel<RAP_DT_F16, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((outproj_ln_h16_kernel<RAP_DT_F16, false>), grid, block, 0, stream, p);
  } else {
    return RAP_ERR_INVALID;
  }
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
