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

#define AQ_WAVE 64
#define AKV 64
#define KLD 68  // K LDS row stride (floats): 272 B = 17 slots -> conflict-free ds_read_b128
#define VLD 64

__global__ __launch_bounds__(256, 2) void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                               int TP, int heads, const AttnWorkItem* __restrict__ items) {
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
  float mrun[2] = {-1e30f, -1e30f};
  float lsum[2] = {0.f, 0.f};

  // ---- K/V staging coordinates: 4 float4 of K and 4 of V per thread per tile
  const int srow = tid >> 4;          // 0..15 (+16*i)
  const int sc4 = (tid & 15) * 4;
  float4 rk0, rk1, rk2, rk3, rv0, rv1, rv2, rv3;
  const int nkv = (len + AKV - 1) / AKV;
  const int koff = srow * KLD + sc4;
  const int voff = srow * VLD + sc4;

#define ATTN_LOAD_ONE(T, I, RK, RV)                                          \
  {                                                                          \
    int key_ = (T) * AKV + srow + 16 * (I);                                  \
    key_ = key_ < len ? key_ : len - 1;                                      \
    RK = *reinterpret_cast<const float4*>(Kg + (size_t)key_ * 64 + sc4);     \
    RV = *reinterpret_cast<const float4*>(Vg + (size_t)key_ * 64 + sc4);     \
  }
#define ATTN_LOAD_TILE(T)          \
  ATTN_LOAD_ONE(T, 0, rk0, rv0)    \
  ATTN_LOAD_ONE(T, 1, rk1, rv1)    \
  ATTN_LOAD_ONE(T, 2, rk2, rv2)    \
  ATTN_LOAD_ONE(T, 3, rk3, rv3)
#define ATTN_STORE_ONE(BUF, I, RK, RV)                                                            \
  *reinterpret_cast<float4*>(&Ks[(BUF) * (AKV * KLD) + koff + 16 * (I) * KLD]) = RK;              \
  *reinterpret_cast<float4*>(&Vs[(BUF) * (AKV * VLD) + voff + 16 * (I) * VLD]) = RV;
#define ATTN_STORE_TILE(BUF)         \
  ATTN_STORE_ONE(BUF, 0, rk0, rv0)   \
  ATTN_STORE_ONE(BUF, 1, rk1, rv1)   \
  ATTN_STORE_ONE(BUF, 2, rk2, rv2)   \
  ATTN_STORE_ONE(BUF, 3, rk3, rv3)

  ATTN_LOAD_TILE(0)
  ATTN_STORE_TILE(0)
  __syncthreads();

  for (int t = 0; t < nkv; ++t) {
    const int cur = t & 1;
    const bool more = (t + 1) < nkv;
    if (more) { ATTN_LOAD_TILE(t + 1) }

    if (wave_active) {
      const float* Kc = Ks + cur * (AKV * KLD);
      const float* Vc = Vs + cur * (AKV * VLD);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const int kbase = t * AKV + kt * 32;
        if (kbase >= len) break;
        // ---- S^T = K Q^T  (32 keys x 2x32 queries)
        f32x16 st[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[0][r] = 0.f; st[1][r] = 0.f; }
        const float* kp = Kc + (kt * 32 + l31) * KLD + 4 * hi;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const float4 kf = *reinterpret_cast<const float4*>(kp + 8 * g);
          const float kv[4] = {kf.x, kf.y, kf.z, kf.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            st[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[j], qf[0][g * 4 + j], st[0], 0, 0, 0);
            st[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[j], qf[1][g * 4 + j], st[1], 0, 0, 0);
          }
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
        // ---- online softmax (lane-local except one cross-half max)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          float mx = st[qt][0];
#pragma unroll
          for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[qt][r]);
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          const float mnew = fmaxf(mrun[qt], mx);
          const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);
          mrun[qt] = mnew;
          float ps = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f(st[qt][r] - mnew);
            st[qt][r] = pv;
            ps += pv;
          }
          lsum[qt] = lsum[qt] * alpha + ps;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            o[qt][0][r] *= alpha;
            o[qt][1][r] *= alpha;
          }
        }
        // ---- O^T += V^T P^T : step r contracts keys {crow(r,0), crow(r,1)} = P register r
        const float* vp = Vc + (kt * 32 + 4 * hi) * VLD + 2 * l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int krow = (r & 3) + 8 * (r >> 2);   // crow(r, 0); the +4*hi is in vp
          const float2 vf = *reinterpret_cast<const float2*>(vp + krow * VLD);
          o[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, st[0][r], o[0][0], 0, 0, 0);
          o[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, st[0][r], o[0][1], 0, 0, 0);
          o[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, st[1][r], o[1][0], 0, 0, 0);
          o[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, st[1][r], o[1][1], 0, 0, 0);
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
    const float ltot = lsum[qt] + __shfl_xor(lsum[qt], 32, 64);
    const float inv = 1.0f / ltot;
    if (q < len) {
      float* op = out + (size_t)(it.seg_start + q) * dmodel + head * 64;
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

// One thread walks the segment table (nseg is small: samples or parts of one batch) and emits one
// work item per 256-query block; unused slots get seg_len = 0.  Runs once per sample() call.
__global__ void build_attn_worklist_kernel(const int32_t* __restrict__ cu, int nseg, AttnWorkItem* __restrict__ items,
                                           int max_items) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int n = 0;
  for (int s = 0; s < nseg; ++s) {
    const int a = cu[s], b = cu[s + 1];
    for (int q0 = 0; q0 < b - a && n < max_items; q0 += RAP_ATTN_BQ) {
      AttnWorkItem w; w.seg_start = a; w.seg_len = b - a; w.q0 = q0; w.pad = 0;
      items[n++] = w;
    }
  }
  for (; n < max_items; ++n) { AttnWorkItem w; w.seg_start = 0; w.seg_len = 0; w.q0 = 0; w.pad = 0; items[n] = w; }
}

int launch_build_attn_worklist(hipStream_t stream, const int32_t* cu_seqlens, int nseg, AttnWorkItem* items,
                               int max_items) {
  if (max_items <= 0) return RAP_OK;
  hipLaunchKernelGGL(build_attn_worklist_kernel, dim3(1), dim3(64), 0, stream, cu_seqlens, nseg, items, max_items);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

int launch_attention_f32(hipStream_t stream, const float* qkv, float* out, int TP, int heads, const AttnWorkItem* items,
                         int max_items) {
  if (max_items <= 0 || TP <= 0) return RAP_OK;
  if (heads <= 0) return RAP_ERR_INVALID;
  hipLaunchKernelGGL(attention_f32_kernel, dim3(max_items * heads), dim3(256), 0, stream, qkv, out, TP, heads, items);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
