// HBM-bound kernels around the velocity network (gfx950): head tail (256 -> 3), Euler update,
// segment tables and the weight re-packing used at model creation.
#include "kernels.h"

// ---------------------------------------------------------------------------------------------
// final_mlp.4 (point_cloud_dit.py:116): v (TP,3) = y (TP,K) * W(3,K)^T, no bias.  One wave per token,
// float4 per lane per 256 columns, three wave reductions.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_out3_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ W,
                                                        float* __restrict__ v, int TP, int K) {
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  for (int tok = wave_global; tok < TP; tok += nwaves) {
    const float* yr = y + (size_t)tok * ldy;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      const float4 yy = *reinterpret_cast<const float4*>(yr + k);
      const float4 w0 = *reinterpret_cast<const float4*>(W + k);
      const float4 w1 = *reinterpret_cast<const float4*>(W + K + k);
      const float4 w2 = *reinterpret_cast<const float4*>(W + 2 * K + k);
      a0 += yy.x * w0.x + yy.y * w0.y + yy.z * w0.z + yy.w * w0.w;
      a1 += yy.x * w1.x + yy.y * w1.y + yy.z * w1.z + yy.w * w1.w;
      a2 += yy.x * w2.x + yy.y * w2.y + yy.z * w2.z + yy.w * w2.w;
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (lane == 0) {
      v[(size_t)tok * 3 + 0] = a0;
      v[(size_t)tok * 3 + 1] = a1;
      v[(size_t)tok * 3 + 2] = a2;
    }
  }
}

int launch_head_out3(hipStream_t stream, const float* y, int ldy, const float* W, float* v, int TP, int K) {
  if (TP <= 0) return RAP_OK;
  if (K % 4 != 0 || ldy % 4 != 0) return RAP_ERR_INVALID;
  int blocks = (TP + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(head_out3_kernel, dim3(blocks), dim3(256), 0, stream, y, ldy, W, v, TP, K);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

// ---------------------------------------------------------------------------------------------
// euler_step (sampler.py:88-90):  x0_hat = x_t - v * t ;  x_t <- x_t - dt * v
// Separate multiply and subtract roundings (no FMA contraction) so the update is bit-identical to the
// reference's tensor ops given the same v.  60 B per point: read x_t, v; write x0_hat (trajectory
// slot), x_t, and the x_t trajectory slot.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void euler_step_kernel(const float* __restrict__ x_t, const float* __restrict__ v, float t,
                                                         float dt, float* __restrict__ x0hat, float* __restrict__ x_next,
                                                         float* __restrict__ traj_xt, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float x = x_t[i], vv = v[i];
    const float x0 = x - mul_rn_nofuse(vv, t);
    const float xn = x - mul_rn_nofuse(dt, vv);
    x0hat[i] = x0;
    x_next[i] = xn;
    if (traj_xt) traj_xt[i] = xn;
  }
}

int launch_euler_step(hipStream_t stream, const float* x_t, const float* v, float t, float dt, float* x0hat_out,
                      float* x_next_out, float* traj_xt_slot_or_null, long n) {
  if (n <= 0) return RAP_OK;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(euler_step_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x_t, v, t, dt, x0hat_out, x_next_out,
                     traj_xt_slot_or_null, n);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

// ---------------------------------------------------------------------------------------------
// segment tables
// ---------------------------------------------------------------------------------------------
// token_sample[t] = b for cu[b] <= t < cu[b+1]  (the device-side equivalent of repeat_by_cu_seqlens,
// utils/point_clouds.py:161-184, without the host sync of repeat_interleave).
__global__ __launch_bounds__(256) void token_sample_kernel(const int32_t* __restrict__ cu, int32_t* __restrict__ ts) {
  const int b = blockIdx.y;
  const int a = cu[b], e = cu[b + 1];
  for (int i = a + blockIdx.x * 256 + threadIdx.x; i < e; i += gridDim.x * 256) ts[i] = b;
}

int launch_token_sample(hipStream_t stream, const int32_t* cu_batch, int B, int32_t* token_sample) {
  if (B <= 0) return RAP_OK;
  hipLaunchKernelGGL(token_sample_kernel, dim3(64, B), dim3(256), 0, stream, cu_batch, token_sample);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

// out[i] = max_{j <= i} clamp(cu[j], 0, limit): the caller's cu_seqlens made safe to index with (under deferred validation
// a malformed table -- last entry beyond the point count, or decreasing -- reached token_sample_kernel and the attention work lists
// un-clamped).  A consistent table is returned unchanged.  One block; thread t owns a contiguous chunk, prefix maximum across chunks.
#define SAN_THREADS 1024
__global__ __launch_bounds__(SAN_THREADS) void sanitize_cu_kernel(const int32_t* __restrict__ cu, int n, int limit, int32_t* __restrict__ out) {
  if (blockIdx.x != 0) return;
  __shared__ int part[SAN_THREADS];
  const int tid = threadIdx.x;
  const int per = (n + SAN_THREADS - 1) / SAN_THREADS;
  const int i0 = tid * per, i1 = (i0 + per) < n ? (i0 + per) : n;
  int m = 0;
  for (int i = i0; i < i1; ++i) { int v = cu[i]; v = v < 0 ? 0 : v > limit ? limit : v; m = v > m ? v : m; }
  part[tid] = m;
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int t = 0; t < SAN_THREADS; ++t) { const int v = part[t]; part[t] = acc; acc = v > acc ? v : acc; } }
  __syncthreads();
  m = part[tid];
  for (int i = i0; i < i1; ++i) { int v = cu[i]; v = v < 0 ? 0 : v > limit ? limit : v; m = v > m ? v : m; out[i] = m; }
}
int launch_sanitize_cu(hipStream_t stream, const int32_t* cu, int n, long limit, int32_t* out) {
  if (n <= 0) return RAP_OK;
  hipLaunchKernelGGL(sanitize_cu_kernel, dim3(1), dim3(SAN_THREADS), 0, stream, cu, n, (int)limit, out);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

// part_offsets[i] = sum_{i' < i} points_per_part.flat[i']   (B*P+1 entries; empty parts have zero length --
// the reference drops them, modeling.py:219-222; zero-length segments are no-ops for every kernel here).
__global__ void part_offsets_kernel(const int64_t* __restrict__ ppp, int nparts, int32_t* __restrict__ off, long limit) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  // limit >= 0: no offset may pass the end of the point arrays (an inconsistent points_per_part then yields short / empty
  // trailing segments -- wrong poses for a malformed batch, but never an out-of-bounds access; rap_check_batch names the defect)
  long acc = 0;
  off[0] = 0;
  for (int i = 0; i < nparts; ++i) {
    const long n = (long)ppp[i];
    acc += n > 0 ? n : 0;
    if (limit >= 0 && acc > limit) acc = limit;
    off[i + 1] = (int)acc;
  }
}

int launch_part_offsets(hipStream_t stream, const int64_t* points_per_part, int nparts, int32_t* part_offsets, long limit) {
  hipLaunchKernelGGL(part_offsets_kernel, dim3(1), dim3(64), 0, stream, points_per_part, nparts, part_offsets, limit);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

// Consistency of a packed batch (what the reference asserts in split_parts, utils/point_clouds.py:33-52): bit 0 sum(points_per_part)
// != TP, bit 1 cu_seqlens[0] != 0 or cu_seqlens[B] != TP, bit 2 cu_seqlens not non-decreasing, bit 3 a sample's parts do not add
// up to its cu_seqlens span, bit 4 a negative part size.
__global__ void check_batch_kernel(const int64_t* __restrict__ ppp, const int32_t* __restrict__ cu, int B, int P, long TP,
                                   int32_t* __restrict__ flag) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int f = 0;
  long total = 0;
  if (cu[0] != 0 || cu[B] != TP) f |= 2;
  for (int b = 0; b < B; ++b) {
    long s = 0;
    for (int p = 0; p < P; ++p) { const long n = (long)ppp[(size_t)b * P + p]; if (n < 0) f |= 16; s += n; }
    total += s;
    if (cu[b + 1] < cu[b]) f |= 4;
    if (s != (long)cu[b + 1] - (long)cu[b]) f |= 8;
  }
  if (total != TP) f |= 1;
  *flag = f;
}

int launch_check_batch(hipStream_t stream, const int64_t* points_per_part, const int32_t* cu_batch, int B, int P, long TP, int32_t* flag) {
  hipLaunchKernelGGL(check_batch_kernel, dim3(1), dim3(64), 0, stream, points_per_part, cu_batch, B, P, TP, flag);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

// Deferred form of the check above (round 4): if *flag != 0 (what check_batch_kernel wrote for this batch), overwrite n floats with
// quiet NaNs -- the results of a sampling call on an inconsistent batch become unmistakable without the host reading anything back.
__global__ __launch_bounds__(256) void poison_on_flag_kernel(const int32_t* __restrict__ flag, float* __restrict__ buf, long n) {
  if (*flag == 0) return;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) buf[i] = __builtin_nanf("");
}
int launch_poison_on_flag(hipStream_t stream, const int32_t* flag, float* buf, long n) {
  if (n <= 0) return RAP_OK;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(poison_on_flag_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, flag, buf, n);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

// ---------------------------------------------------------------------------------------------
// weight re-packing (model creation only)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ src, int src_ld, int src_col0,
                                                        float* __restrict__ dst, int dst_ld, int dst_col0, int rows, int cols) {
  const long n = (long)rows * cols;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    dst[(size_t)r * dst_ld + dst_col0 + c] = src[(size_t)r * src_ld + src_col0 + c];
  }
}

int launch_copy_cols(hipStream_t stream, const float* src, int src_ld, int src_col0, float* dst, int dst_ld, int dst_col0,
                     int rows, int cols) {
  const long n = (long)rows * cols;
  if (n <= 0) return RAP_OK;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, src_ld, src_col0, dst, dst_ld, dst_col0,
                     rows, cols);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

// GEGLU projection (diffusers GEGLU: proj rows [0,inner) = value, [inner, 2*inner) = gate): interleave in
// groups of 32 so that one 64-column wave tile of the GEMM holds value and gate of the same 32 outputs:
//   Wp[64*g + c]      = W[32*g + c]            (c < 32)
//   Wp[64*g + 32 + c] = W[inner + 32*g + c]
__global__ __launch_bounds__(256) void geglu_interleave_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                               float* __restrict__ Wp, float* __restrict__ bp, int inner, int K) {
  const int rp = blockIdx.x;                 // permuted row in [0, 2*inner)
  const int g = rp >> 6, c = rp & 63;
  const int src = (c < 32) ? (32 * g + c) : (inner + 32 * g + (c - 32));
  for (int k = threadIdx.x; k < K; k += 256) Wp[(size_t)rp * K + k] = W[(size_t)src * K + k];
  if (threadIdx.x == 0) bp[rp] = b[src];
}

int launch_geglu_interleave(hipStream_t stream, const float* W, const float* b, float* Wp, float* bp, int inner, int K) {
  if (inner % 32 != 0) return RAP_ERR_INVALID;
  hipLaunchKernelGGL(geglu_interleave_kernel, dim3(2 * inner), dim3(256), 0, stream, W, b, Wp, bp, inner, K);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

__global__ __launch_bounds__(256) void fill_zero_kernel(float* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0.f;
}

int launch_fill_zero(hipStream_t stream, float* p, size_t n) {
  if (n == 0) return RAP_OK;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, n);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
