// Variable-length, non-causal softmax attention in fp32 on the CDNA4 matrix cores.
//
// Replaces flash_attn.flash_attn_varlen_qkvpacked_func as called by the reference at
// flow_model/layer.py:106-111 (per part) and :123-128 (per sample): for every segment of
// cu_seqlens and every head,  out = softmax(q k^T / sqrt(64)) v  with no attention across segments.
//
// Layout: q/k/v are head-major [3][H][TP][64] fp32 (written by the QKV GEMM epilogue), so each
// (segment, head) problem is three dense (L,64) matrices; out is token-major (TP, H*64).
//
// Design (gfx950 only; Dh = 64):
//  * flash-style online softmax; a block = 256 query rows of one (segment, head) -- 4 waves x 64 rows;
//    K/V are streamed in 64-key tiles through double-buffered LDS (global -> regs -> LDS).
//  * "swapped" products on v_mfma_f32_32x32x2_f32 (exact fp32):
//      S^T (key x query) = K (key x d) * Q^T (d x query)    -- A operand from LDS, B operand = Q in VGPRs
//      O^T (d x query)   = V^T (d x key) * P^T (key x query) -- A operand from LDS, B operand = P in VGPRs
//    In the 32x32 C/D layout a lane then owns ONE query column (lane & 31) and 16 of the 32 keys, so the
//    whole softmax state (running max, running sum, rescale of O) is lane-local; one cross-half
//    exchange (lane ^ 32) per 32-key tile completes the row max.
//  * No data movement between the two products: MFMA step r of P*V contracts the key pair
//    {crow(r,0), crow(r,1)} = exactly the keys whose probabilities already sit in accumulator register r
//    of the two half-waves, so P's registers are fed straight back as the B operand.
//  * d is permuted between the two O tiles (tile e holds d = 2c + e) so that one ds_read_b64 of V
//    feeds both; the permutation is undone in the store.
//  * Contraction order over d in Q*K^T is permuted as in the GEMM (one ds_read_b128 -> four MFMAs).
//  * softmax scale and log2(e) are folded into Q when it is loaded; exponentials are v_exp_f32.
//  * grid = work items x heads with head = blockIdx % H (blockIdx % 8 is the XCD on MI355X: each XCD's
//    L2 then serves the K/V of one head).
#include "kernels.h"

// Measured and NOT kept (round 3, call 22): K / V tiles by LDS-DMA (global_load_lds_dwordx4, 256-byte K rows with a slot ^ (row & 15) swizzle)
// instead of through staging registers -- bit-identical output, 196 instead of 228-242 VGPRs, and exactly the same speed (141.7 vs
// 141.8 TF per sample, 140.8 vs 141.2 per part): at 64 cycles per MFMA the 8 loads + 8 ds_write_b128 of a tile are noise.  The same
// change is worth +3.5 % in the 16-bit kernel (attn_h16.hip), where a tile is 16x shorter.
#define AQ_WAVE 64
#define AKV 64
#define KLD 68  // K LDS row stride (floats): 272 B = 17 slots -> conflict-free ds_read_b128
#define VLD 64
// scheduling fence: VALU (0x2), SALU (0x4) and transcendental (0x400) instructions may cross; DS, MFMA, VMEM may not
#define ATTN_FENCE __builtin_amdgcn_sched_barrier(0x406);

__device__ __forceinline__ float xhalf_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

#define ATTN_DEFER_THR 11.5f   // = 8 in natural-log units: the online softmax rescales only when a row maximum grows by more than e^8
__device__ __forceinline__ float amax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// NW = waves per block (4: 256 queries, two blocks per CU; 8: 512 queries, one block per CU).  With NW = 8 the
// two waves that share a SIMD (w and w+4) belong to the SAME block and get different static priorities: with equal
// priority two identical waves share the matrix pipe fairly, stay in lock-step and hit their softmax (VALU) phases
// together, leaving the pipe idle; with a priority split the favoured wave runs its MFMA phases at full rate and
// the other one fills its gaps (MI355X_MICROARCH.md "Two waves per SIMD", item 4).
// BOUNDED: the caller guarantees q.k/8 <= bound[head] for every pair (after the reference's qk-norm
// bound = 8 max|gamma_q| max|gamma_k|, a property of the weights).  The softmax is then evaluated with that constant as
// its offset, p = exp(s - bound): no running maximum, no rescale of O, ~1/3 of the VALU work between the two MFMA phases
// (the section that costs the online version its last 13 % of the matrix peak).  Same softmax, different rounding points.
// SPLIT (few-token calls, BOUNDED only): gridDim.y key ranges per (work item, head).  With the offset-free softmax the partial
// results of disjoint key ranges simply ADD (numerators and row sums share the common factor 1), so every range writes its
// un-normalised O and l to a scratch plane and attention_combine_kernel divides the sums: 4x the blocks for a call whose work
// list would otherwise cover a quarter of the CUs.
template <int NW, bool BOUNDED, bool SPLIT = false>
__global__ __launch_bounds__(64 * NW, 2) void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                  int TP, int heads, const AttnWorkItem* __restrict__ items,
                                                                  const float* __restrict__ bound, float* __restrict__ part_o,
                                                                  float* __restrict__ part_l) {
  __shared__ __attribute__((aligned(16))) float smem[2 * AKV * KLD + 2 * AKV * VLD];
  float* Ks = smem;                    // [2][64][68]
  float* Vs = smem + 2 * AKV * KLD;    // [2][64][64]

  const int head = blockIdx.x % heads;
  const AttnWorkItem it = items[blockIdx.x / heads];
  const int len = it.seg_len;
  if (len <= 0) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;

  const size_t plane = (size_t)heads * TP * 64;
  const float* Qg = qkv + ((size_t)head * TP + it.seg_start) * 64;
  const float* Kg = Qg + plane;
  const float* Vg = Qg + 2 * plane;

  const int qw0 = it.q0 + wave * AQ_WAVE;
  const bool wave_active = qw0 < len;   // waves beyond the segment still help stage K/V
  if (NW == 8) {
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
  }

  // ---- Q fragments: qf[qt][g*4+j] = Q[q][8g + 4hi + j] * (log2(e)/8)
  const float qscale = 0.125f * 1.44269504088896340736f;
  float qf[2][32];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    int q = qw0 + qt * 32 + l31;
    q = q < len ? q : len - 1;
    const float* qp = Qg + (size_t)q * 64 + 4 * hi;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const float4 v = *reinterpret_cast<const float4*>(qp + 8 * g);
      qf[qt][g * 4 + 0] = v.x * qscale;
      qf[qt][g * 4 + 1] = v.y * qscale;
      qf[qt][g * 4 + 2] = v.z * qscale;
      qf[qt][g * 4 + 3] = v.w * qscale;
    }
  }

  f32x16 o[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[a][b][r] = 0.f;
  // BOUNDED needs bound <= 40 (|score| * log2(e) <= 58, see the softmax section); a larger bound poisons this head's output with
  // NaN instead of overflowing silently (rap_model_create only selects this kernel when every bound passes).
  const bool bound_ok = BOUNDED ? (bound[head] <= 40.0f) : true;
  float mrun[2] = {-1e30f, -1e30f};
  float lsum[2] = {0.f, 0.f};

  // ---- K/V staging coordinates: 4 float4 of K and 4 of V per thread per tile
  constexpr int RS = 4 * NW;          // rows staged per pass (16 threads per 64-float row)
  const int srow = tid >> 4;          // 0..RS-1 (+RS*i)
  const int sc4 = (tid & 15) * 4;
  float4 rk0, rk1, rk2, rk3, rv0, rv1, rv2, rv3;
  const int nkv = (len + AKV - 1) / AKV;
  int t_begin = 0, t_end = nkv;
  if (SPLIT) {
    t_begin = (int)((long)nkv * blockIdx.y / gridDim.y);
    t_end = (int)((long)nkv * (blockIdx.y + 1) / gridDim.y);
  }
  const int koff = srow * KLD + sc4;
  const int voff = srow * VLD + sc4;

#define ATTN_LOAD_ONE(T, I, RK, RV)                                          \
  {                                                                          \
    int key_ = (T) * AKV + srow + RS * (I);                                  \
    key_ = key_ < len ? key_ : len - 1;                                      \
    RK = *reinterpret_cast<const float4*>(Kg + (size_t)key_ * 64 + sc4);     \
    RV = *reinterpret_cast<const float4*>(Vg + (size_t)key_ * 64 + sc4);     \
  }
#define ATTN_LOAD_TILE(T)          \
  ATTN_LOAD_ONE(T, 0, rk0, rv0)    \
  ATTN_LOAD_ONE(T, 1, rk1, rv1)    \
  if constexpr (NW == 4) {         \
    ATTN_LOAD_ONE(T, 2, rk2, rv2)  \
    ATTN_LOAD_ONE(T, 3, rk3, rv3)  \
  }
#define ATTN_STORE_ONE(BUF, I, RK, RV)                                                            \
  *reinterpret_cast<float4*>(&Ks[(BUF) * (AKV * KLD) + koff + RS * (I) * KLD]) = RK;              \
  *reinterpret_cast<float4*>(&Vs[(BUF) * (AKV * VLD) + voff + RS * (I) * VLD]) = RV;
#define ATTN_STORE_TILE(BUF)           \
  ATTN_STORE_ONE(BUF, 0, rk0, rv0)     \
  ATTN_STORE_ONE(BUF, 1, rk1, rv1)     \
  if constexpr (NW == 4) {             \
    ATTN_STORE_ONE(BUF, 2, rk2, rv2)   \
    ATTN_STORE_ONE(BUF, 3, rk3, rv3)   \
  }

  ATTN_LOAD_TILE(t_begin)
  ATTN_STORE_TILE(0)
  __syncthreads();

  for (int t = t_begin; t < t_end; ++t) {
    const int cur = (t - t_begin) & 1;
    const bool more = (t + 1) < t_end;
    if (more) { ATTN_LOAD_TILE(t + 1) }

    if (wave_active) {
      const float* Kc = Ks + cur * (AKV * KLD);
      const float* Vc = Vs + cur * (AKV * VLD);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const int kbase = t * AKV + kt * 32;
        if (kbase >= len) break;
        // ---- S^T = K Q^T  (32 keys x 2x32 queries).  LDS fragments are fetched one group ahead of the MFMAs
        // that consume them and fenced in place (ATTN_FENCE lets VALU/SALU cross, pins DS and MFMA): hipcc
        // otherwise sinks each ds_read to just before its consumer and ~130 cycles of LDS latency are exposed per
        // 8 MFMAs (v1.0 of this kernel: 86 % of the fp32 matrix peak).
        f32x16 st[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[0][r] = 0.f; st[1][r] = 0.f; }
        const float* kp = Kc + (kt * 32 + l31) * KLD + 4 * hi;
        const float* vp = Vc + (kt * 32 + 4 * hi) * VLD + 2 * l31;
        float4 kf = *reinterpret_cast<const float4*>(kp);
        float2 vfa, vfb;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          float4 kn = kf;
          if (g < 7) kn = *reinterpret_cast<const float4*>(kp + 8 * (g + 1));
          else { vfa = *reinterpret_cast<const float2*>(vp); vfb = *reinterpret_cast<const float2*>(vp + VLD); }
          ATTN_FENCE
          st[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[0][g * 4 + 0], st[0], 0, 0, 0);
          st[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[1][g * 4 + 0], st[1], 0, 0, 0);
          st[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[0][g * 4 + 1], st[0], 0, 0, 0);
          st[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[1][g * 4 + 1], st[1], 0, 0, 0);
          st[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[0][g * 4 + 2], st[0], 0, 0, 0);
          st[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[1][g * 4 + 2], st[1], 0, 0, 0);
          st[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[0][g * 4 + 3], st[0], 0, 0, 0);
          st[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[1][g * 4 + 3], st[1], 0, 0, 0);
          ATTN_FENCE
          kf = kn;
        }
        // ---- mask keys beyond the segment (only the last tile can be partial)
        if (kbase + 32 > len) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool valid = (kbase + mfma32_crow(r, hi)) < len;
            st[0][r] = valid ? st[0][r] : -1e30f;
            st[1][r] = valid ? st[1][r] : -1e30f;
          }
        }
        if (BOUNDED) {
          // ---- softmax numerators with NO offset: |score| <= bound * log2(e) <= 58 (the launcher checks bound <= 40), so
          // exp2(score) lies in [2^-58, 2^58] and every sum stays far inside fp32; the common factor cancels in O / l.
          // Row sums through two independent chains.
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
            float pa = 0.f, pb2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const float e0 = __builtin_amdgcn_exp2f(st[qt][r]);
              const float e1 = __builtin_amdgcn_exp2f(st[qt][r + 1]);
              st[qt][r] = e0; st[qt][r + 1] = e1;
              pa += e0; pb2 += e1;
            }
            lsum[qt] += pa + pb2;
          }
        } else {
        // ---- online softmax (lane-local except one cross-half max, a VALU v_permlane32_swap).  Round 3: row maximum through
        // v_max3_f32 (8 instead of 15 instructions per 16 scores; hipcc also prepends a canonicalising v_max to every
        // MFMA-output operand of fmaxf) and a DEFERRED rescale, as the 16-bit kernel has had since r01: the running reference
        // m of a row moves only when a row of the wave outgrows it by more than 2^11.5 (e^8), so in the steady state a
        // sub-tile costs no alpha, no 32 multiplies of O and no l * alpha -- p = 2^(s - m) <= 2^11.5 and every sum stays far
        // inside fp32; softmax is invariant to the reference.  Same function, different rounding pattern.
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          float mx = amax3(st[qt][0], st[qt][1], st[qt][2]);
#pragma unroll
          for (int r = 3; r < 15; r += 2) mx = amax3(mx, st[qt][r], st[qt][r + 1]);
          mx = xhalf_max(fmaxf(mx, st[qt][15]));
          if (__builtin_amdgcn_ballot_w64(mx > mrun[qt] + ATTN_DEFER_THR) != 0) {       // wave-uniform
            const float mnew = fmaxf(mrun[qt], mx);
            const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);
            mrun[qt] = mnew;
            lsum[qt] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              o[qt][0][r] *= alpha;
              o[qt][1][r] *= alpha;
            }
          }
          float pa = 0.f, pb2 = 0.f;
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const float e0 = __builtin_amdgcn_exp2f(st[qt][r] - mrun[qt]);
            const float e1 = __builtin_amdgcn_exp2f(st[qt][r + 1] - mrun[qt]);
            st[qt][r] = e0; st[qt][r + 1] = e1;
            pa += e0; pb2 += e1;
          }
          lsum[qt] += pa + pb2;
        }
        }
        // ---- O^T += V^T P^T : step r contracts keys {crow(r,0), crow(r,1)} = P register r.  V fragments are
        // fetched two steps (8 MFMAs) ahead.
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float2 vna = vfa, vnb = vfb;
          if (r + 2 < 16) {
            const int k0 = ((r + 2) & 3) + 8 * ((r + 2) >> 2);   // crow(r+2, 0); the +4*hi is in vp
            const int k1 = ((r + 3) & 3) + 8 * ((r + 3) >> 2);
            vna = *reinterpret_cast<const float2*>(vp + k0 * VLD);
            vnb = *reinterpret_cast<const float2*>(vp + k1 * VLD);
          }
          ATTN_FENCE
          o[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vfa.x, st[0][r], o[0][0], 0, 0, 0);
          o[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vfa.y, st[0][r], o[0][1], 0, 0, 0);
          o[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vfa.x, st[1][r], o[1][0], 0, 0, 0);
          o[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vfa.y, st[1][r], o[1][1], 0, 0, 0);
          o[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vfb.x, st[0][r + 1], o[0][0], 0, 0, 0);
          o[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vfb.y, st[0][r + 1], o[0][1], 0, 0, 0);
          o[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vfb.x, st[1][r + 1], o[1][0], 0, 0, 0);
          o[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vfb.y, st[1][r + 1], o[1][1], 0, 0, 0);
          ATTN_FENCE
          vfa = vna; vfb = vnb;
        }
      }
    }

    if (more) { ATTN_STORE_TILE(cur ^ 1) }
    __syncthreads();
  }

  if (!wave_active) return;
  // ---- normalise and store: lane owns query (lane&31) of each q-tile; register r of tile e is d = 2*crow(r,hi)+e
  const int dmodel = heads * 64;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = qw0 + qt * 32 + l31;
    const float ltot = xhalf_sum(lsum[qt]);
    const float inv = SPLIT ? 1.0f : (bound_ok ? 1.0f / ltot : __builtin_nanf(""));
    if (q < len) {
      float* op = (SPLIT ? part_o + (size_t)blockIdx.y * TP * dmodel : out) + (size_t)(it.seg_start + q) * dmodel + head * 64;
      if (SPLIT && hi == 0)
        part_l[((size_t)blockIdx.y * TP + it.seg_start + q) * heads + head] = bound_ok ? ltot : __builtin_nanf("");
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        // registers 4rg..4rg+3 -> c = 8rg + 4hi + (0..3) -> d = 2c .. 2c+7 : 8 contiguous floats
        const int d0 = 2 * (8 * rg + 4 * hi);
        float4 w0, w1;
        w0.x = o[qt][0][4 * rg + 0] * inv; w0.y = o[qt][1][4 * rg + 0] * inv;
        w0.z = o[qt][0][4 * rg + 1] * inv; w0.w = o[qt][1][4 * rg + 1] * inv;
        w1.x = o[qt][0][4 * rg + 2] * inv; w1.y = o[qt][1][4 * rg + 2] * inv;
        w1.z = o[qt][0][4 * rg + 3] * inv; w1.w = o[qt][1][4 * rg + 3] * inv;
        *reinterpret_cast<float4*>(op + d0) = w0;
        *reinterpret_cast<float4*>(op + d0 + 4) = w1;
      }
    }
  }
}

// Work list of one attention launch: one item per 256-query block of every segment, LONGEST SEGMENT FIRST (round 4).  A block's
// work is proportional to its segment's length (it streams all keys of the segment), and blocks are dispatched in blockIdx order:
// with the items in segment order the 157 x 8 blocks of a 40 000-point part at the end of a ragged batch start last and the launch
// drains on them alone; longest-processing-time-first puts the short blocks at the tail (the reference's batches are ragged:
// 200 ... 40 000 points per part, config/RAP_inference.yaml:30-36).  Items of one segment stay adjacent, so the 8 blocks x
// consecutive items that share an XCD's L2 still share K / V.  One block of 1024 threads, once per sample() call:
//   rank[s] = #{j : len_j > len_s or (len_j == len_s and j < s)}  (lengths staged through LDS, O(nseg^2 / 1024) per thread),
//   order[rank[s]] = s, exclusive scan of the item counts in rank order, then every thread emits the items of its ranks.
// sort_ws: nseg ints; nullptr = segment order (kernel-level entry points without scratch).
#define WL_THREADS 1024
#define WL_TILE 4096
#define WL_SORT_MAX 8192
// a segment's length as every stage of the kernel sees it: never negative (a decreasing table -- reachable through the kernel-level
// entry point, or under deferred validation before round 5 sanitised the tables -- made the item count negative)
__device__ __forceinline__ int wl_len(const int32_t* __restrict__ cu, int s) { const int l = cu[s + 1] - cu[s]; return l > 0 ? l : 0; }
__global__ __launch_bounds__(WL_THREADS) void build_attn_worklist_kernel(const int32_t* __restrict__ cu, int nseg, AttnWorkItem* __restrict__ items,
                                                                          int max_items, int bq, int32_t* __restrict__ sort_ws) {
  if (blockIdx.x != 0) return;
  const int tid = threadIdx.x;
  const AttnWorkItem none = {0, 0, 0, 0};
  if (!sort_ws) {
    if (tid != 0) return;
    int n = 0;
    for (int s = 0; s < nseg; ++s) {
      const int a = cu[s], len = wl_len(cu, s);
      for (int q0 = 0; q0 < len && n < max_items; q0 += bq) { const AttnWorkItem w = {a, len, q0, 0}; items[n++] = w; }
    }
    for (; n < max_items; ++n) items[n] = none;
    return;
  }
  __shared__ int lens[WL_TILE];
  __shared__ int partial[WL_THREADS];
  __shared__ int total_items;
  int32_t* order = sort_ws;            // [nseg]
  // ---- ranks.  O(nseg^2 / 1024) compares per thread on ONE CU: fine for the hundreds of parts of a real batch, ~4 M per thread at the
  // 65 535-part limit (milliseconds at the head of every call) -- above WL_SORT_MAX segments the items are emitted in
  // segment order by the same parallel scan (such batches consist of many short segments: there is no long tail to hide)
  if (nseg > WL_SORT_MAX) {
    for (int s = tid; s < nseg; s += WL_THREADS) order[s] = s;
  } else
  for (int s0 = 0; s0 < nseg; s0 += WL_THREADS) {          // segments owned by this thread in this pass: s0 + tid
    const int s = s0 + tid;
    const int my = s < nseg ? wl_len(cu, s) : -1;
    int rank = 0;
    for (int j0 = 0; j0 < nseg; j0 += WL_TILE) {
      __syncthreads();
      for (int j = tid; j < WL_TILE && j0 + j < nseg; j += WL_THREADS) lens[j] = wl_len(cu, j0 + j);
      __syncthreads();
      const int nj = nseg - j0 < WL_TILE ? nseg - j0 : WL_TILE;
      if (s < nseg)
        for (int j = 0; j < nj; ++j) rank += (lens[j] > my || (lens[j] == my && j0 + j < s)) ? 1 : 0;
    }
    if (s < nseg) order[rank] = s;
  }
  __threadfence_block();
  __syncthreads();
  // ---- exclusive scan of the item counts in rank order: thread t owns ranks [t * per, (t + 1) * per)
  const int per = (nseg + WL_THREADS - 1) / WL_THREADS;
  const int r0 = tid * per, r1 = (r0 + per) < nseg ? (r0 + per) : nseg;
  int sum = 0;
  for (int r = r0; r < r1; ++r) { const int sg = order[r]; sum += (wl_len(cu, sg) + bq - 1) / bq; }
  partial[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int t = 0; t < WL_THREADS; ++t) { const int v = partial[t]; partial[t] = acc; acc += v; }
    total_items = acc;
  }
  __syncthreads();
  // ---- emit
  int n = partial[tid];
  for (int r = r0; r < r1; ++r) {
    const int sg = order[r];
    const int a = cu[sg], len = wl_len(cu, sg);
    for (int q0 = 0; q0 < len; q0 += bq, ++n)
      if (n < max_items) { const AttnWorkItem w = {a, len, q0, 0}; items[n] = w; }
  }
  __syncthreads();                     // the padding below must not race with another thread's emit
  const int total = total_items < 0 ? 0 : total_items < max_items ? total_items : max_items;
  for (int i = total + tid; i < max_items; i += WL_THREADS) items[i] = none;
}

// One schedule ships: 4 waves x 64 queries per block (136 -> 142 TF at the C1 shapes).  The 8-wave / 512-query form (134 TF) and
// the software-pipelined v3 kernel (124 TF) of round 1 are in the history at 72efb73 (profiles/r01_run3_kernel_variant_sweep.jsonl).
rap_tuning_t g_rap_attn_variant = 1;   // kept for the ablation build's rap_set_tuning(1, .): only 1 exists
rap_tuning_t g_rap_attn_split = 1;     // tuning key 5: 0 = never split few-token calls over key ranges
static int attn_block_queries() { return RAP_ATTN_BQ; }

int launch_build_attn_worklist(hipStream_t stream, const int32_t* cu_seqlens, int nseg, AttnWorkItem* items,
                               int max_items, int block_queries, int32_t* sort_ws) {
  if (max_items <= 0) return RAP_OK;
  hipLaunchKernelGGL(build_attn_worklist_kernel, dim3(1), dim3(WL_THREADS), 0, stream, cu_seqlens, nseg, items, max_items,
                     block_queries > 0 ? block_queries : attn_block_queries(), sort_ws);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

// out[t][c] = sum_s part_o[s][t][c] / sum_s part_l[s][t][head(c)]   (one thread per 4 columns)
__global__ __launch_bounds__(256) void attention_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_l,
                                                                float* __restrict__ out, int TP, int heads, int splits,
                                                                const int32_t* __restrict__ cover_cu, int cover_nseg) {
  const int dmodel = heads * 64;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;           // float4 index
  if (i >= (long)TP * dmodel / 4) return;
  const long t = i / (dmodel / 4);
  const int c = (int)(i % (dmodel / 4)) * 4;
  if (cover_cu && (t < cover_cu[0] || t >= cover_cu[cover_nseg])) {      // no block wrote partials for this row (filler rows, rows outside every segment)
    *reinterpret_cast<float4*>(out + t * dmodel + c) = float4{0.f, 0.f, 0.f, 0.f};
    return;
  }
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  float l = 0.f;
  for (int s = 0; s < splits; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(part_o + ((size_t)s * TP + t) * dmodel + c);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    l += part_l[((size_t)s * TP + t) * heads + (c >> 6)];
  }
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
  *reinterpret_cast<float4*>(out + t * dmodel + c) = acc;
}

// How many key ranges a few-token call is split into (1 = no split): only the bounded (offset-free) kernel can add partial
// results, and only work lists that leave most of the 2 x 256 block slots empty are worth the extra pass.
int attention_f32_splits(int max_items, int heads, bool bounded) {
  if (!bounded || g_rap_attn_split == 0) return 1;
  const long blocks = (long)max_items * heads;
  return blocks <= 160 ? 4 : blocks <= 320 ? 2 : 1;
}

int launch_attention_f32(hipStream_t stream, const float* qkv, float* out, int TP, int heads, const AttnWorkItem* items,
                         int max_items, const float* bound, float* part_o, float* part_l, int splits, const int32_t* cover_cu,
                         int cover_nseg) {
  if (max_items <= 0 || TP <= 0) return RAP_OK;
  if (heads <= 0) return RAP_ERR_INVALID;
  if (splits > 1) {
    if (!bound || !part_o || !part_l) return RAP_ERR_INVALID;
    // At most ~one block per CU in the grid (one pair of 2 x 1024 points: 8 work items x 8 heads x 4 key ranges = 256 blocks): ask for
    // 16 KB of (unused) dynamic LDS on top of the 68 KB of tiles, so that only ONE block fits a CU.  Left to itself the dispatcher
    // doubles blocks up on some CUs while others stay empty, and two 4-wave blocks on one CU share its four matrix pipes: 102 -> 72 us
    // per launch, 62.5 -> 55.3 ms per fp32 call at that size (r03 call 36; the same trick changes nothing for the 16-bit attention
    // and GEMM launches of such a call).
    unsigned dyn = 0;
    if ((long)max_items * heads * splits <= 384) {
      dyn = 16384;
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_f32_kernel<4, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, dyn) != hipSuccess) {
        rap_set_last_hip_error((int)hipGetLastError());
        return RAP_ERR_HIP;
      }
    }
    hipLaunchKernelGGL((attention_f32_kernel<4, true, true>), dim3(max_items * heads, splits), dim3(256), dyn, stream, qkv, out, TP,
                       heads, items, bound, part_o, part_l);
    RAP_LAUNCH_CHECK();
    const long n4 = (long)TP * heads * 16;
    hipLaunchKernelGGL(attention_combine_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, part_o, part_l, out, TP,
                       heads, splits, cover_cu, cover_nseg);
    RAP_LAUNCH_CHECK();
    return RAP_OK;
  }
  float* const no = nullptr;
  if (bound) hipLaunchKernelGGL((attention_f32_kernel<4, true>), dim3(max_items * heads), dim3(256), 0, stream, qkv, out, TP, heads, items, bound, no, no);
  else hipLaunchKernelGGL((attention_f32_kernel<4, false>), dim3(max_items * heads), dim3(256), 0, stream, qkv, out, TP, heads, items, bound, no, no);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
