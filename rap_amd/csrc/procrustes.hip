// Per-part Procrustes / Kabsch on device (gfx950): replaces the Python double loop of
// rectified_point_flow/procrustes.py (solve_procrustes :6-37, fit_transformations :40-84,
// rigidify_prediction_with_procrustes :86-118), which costs one torch.linalg.svd launch and >= 2
// device->host syncs per part per flow step.
//
//   mu_s = mean(src_p), mu_t = mean(tgt_p), H = (src_p - mu_s)^T (tgt_p - mu_t)          (:20-26)
//   U S V^T = svd(H);  R = V U^T;  det R < 0 -> flip the row of V^T of the smallest sigma    (:27-33)
//   t = mu_t - mu_s R^T ;  apply as  x R^T + t                                              (:36, :114)
//
// Three kernels, no host involvement:
//  1. procrustes_moments: grid (<= 16 chunks, parts), ~2048 points per chunk block (8 per thread); every block reduces 15 raw moments
//     of its chunk (sum s, sum t, sum s t^T) in fp64 with ONE reduce-scatter butterfly per wave and writes one partial record --
//     24 B read per point, deterministic (no atomics).
//  2. procrustes_solve: one lane per part sums the part's partials in fixed order, forms the centred H
//     in fp64, runs a one-sided Jacobi SVD of the 3x3 and writes R (row-major) and t in fp32.
//     Empty parts produce all-zero R, t (the reference leaves zero rows, procrustes.py:71-76).
//  3. rigid_apply: out = src R^T + t, optionally blended x_t = out*w0 + x_1*w1 (sampler.py:60) and
//     mirrored into the trajectory slot.
#include "kernels.h"
#include "kabsch.h"

// chunks a part of n points is split over: ~2048 points (8 per thread) per block, at most RAP_PROC_CHUNKS.  Round 6: the grid used to
// run all 16 chunk blocks for every part -- 1 point per thread at configs[1] (4096-point parts), i.e. 1024 blocks whose time was the 15
// fp64 wave reductions, not the 24 B per point they read (r05: 11.1 us for 6.3 MB).  Blocks beyond a part's chunk count leave at once.
__device__ __forceinline__ int proc_chunks_of(int n) {
  const int c = (n + 2047) / 2048;
  return c < 1 ? 1 : c > RAP_PROC_CHUNKS ? RAP_PROC_CHUNKS : c;
}

// Sum of 16 per-lane doubles over the 64 lanes of a wave by a reduce-scatter butterfly: at distance 32 / 16 / 8 / 4 a lane hands the
// half of its values it does not keep to its partner (8 + 4 + 2 + 1 exchanges), the last value is summed over distances 2 and 1 --
// 17 fp64 exchanges instead of the 15 x 6 = 90 of one plain butterfly per moment.  On return lane L with L % 4 == 0 holds the total of
// value index k = bit5(L) * 8 + bit4(L) * 4 + bit3(L) * 2 + bit2(L).  Fixed association: deterministic.
__device__ __forceinline__ double wave_sum16_scatter(double (&v)[16], int lane) {
  double w8[8], w4[4], w2[2];
  {
    const bool hi = (lane & 32) != 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double keep = hi ? v[8 + j] : v[j], give = hi ? v[j] : v[8 + j];
      w8[j] = keep + __shfl_xor(give, 32, 64);
    }
  }
  {
    const bool hi = (lane & 16) != 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double keep = hi ? w8[4 + j] : w8[j], give = hi ? w8[j] : w8[4 + j];
      w4[j] = keep + __shfl_xor(give, 16, 64);
    }
  }
  {
    const bool hi = (lane & 8) != 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const double keep = hi ? w4[2 + j] : w4[j], give = hi ? w4[j] : w4[2 + j];
      w2[j] = keep + __shfl_xor(give, 8, 64);
    }
  }
  double r;
  {
    const bool hi = (lane & 4) != 0;
    const double keep = hi ? w2[1] : w2[0], give = hi ? w2[0] : w2[1];
    r = keep + __shfl_xor(give, 4, 64);
  }
  r += __shfl_xor(r, 2, 64);
  r += __shfl_xor(r, 1, 64);
  return r;
}

struct __attribute__((packed, aligned(4))) P3 { float x, y, z; };

__global__ __launch_bounds__(256) void procrustes_moments_kernel(const float* __restrict__ src, const float* __restrict__ tgt,
                                                                 const int32_t* __restrict__ off, double* __restrict__ partials) {
  __shared__ double red[4][16];
  const int part = blockIdx.y, chunk = blockIdx.x;
  const int a = off[part], n = off[part + 1] - a;
  const int nc = proc_chunks_of(n);
  if (chunk >= nc) return;
  const long lo = a + (long)n * chunk / nc;
  const long hi = a + (long)n * (chunk + 1) / nc;
  double m[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) m[k] = 0.0;
  // up to 8 points per thread in flight at once (all 48 loads of a 2048-point batch are requested before the first is used: a loop of
  // one point per iteration pays one memory round trip per point); points beyond the chunk are clamped to its last point and weighted 0
  for (long base = lo; base < hi; base += 2048) {
    float sv[8][3], tv[8][3];
    double wgt[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      long i = base + threadIdx.x + u * 256;
      wgt[u] = i < hi ? 1.0 : 0.0;
      i = i < hi ? i : hi - 1;
      // one 12-byte load per point (global_load_dwordx3 needs 4-byte alignment only) instead of three strided dword loads
      const P3 ps = reinterpret_cast<const P3*>(src)[i], pt = reinterpret_cast<const P3*>(tgt)[i];
      sv[u][0] = ps.x; sv[u][1] = ps.y; sv[u][2] = ps.z; tv[u][0] = pt.x; tv[u][1] = pt.y; tv[u][2] = pt.z;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double s0 = sv[u][0] * wgt[u], s1 = sv[u][1] * wgt[u], s2 = sv[u][2] * wgt[u];
      const double t0 = tv[u][0] * wgt[u], t1 = tv[u][1] * wgt[u], t2 = tv[u][2] * wgt[u];
      m[0] += s0; m[1] += s1; m[2] += s2;
      m[3] += t0; m[4] += t1; m[5] += t2;
      m[6] += s0 * t0; m[7] += s0 * t1; m[8] += s0 * t2;
      m[9] += s1 * t0; m[10] += s1 * t1; m[11] += s1 * t2;
      m[12] += s2 * t0; m[13] += s2 * t1; m[14] += s2 * t2;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double tot = wave_sum16_scatter(m, lane);
  if ((lane & 3) == 0) red[wave][(((lane >> 5) & 1) << 3) | (((lane >> 4) & 1) << 2) | (((lane >> 3) & 1) << 1) | ((lane >> 2) & 1)] = tot;
  __syncthreads();
  if (threadIdx.x < 15) {
    const int k = threadIdx.x;
    partials[((size_t)part * RAP_PROC_CHUNKS + chunk) * 16 + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
  }
}

__global__ __launch_bounds__(64) void procrustes_solve_kernel(const double* __restrict__ partials, const int32_t* __restrict__ off,
                                                              int nparts, float* __restrict__ R_out, float* __restrict__ t_out) {
  const int part = blockIdx.x * 64 + threadIdx.x;
  if (part >= nparts) return;
  const int n = off[part + 1] - off[part];
  float* Ro = R_out + (size_t)part * 9;
  float* to = t_out + (size_t)part * 3;
  if (n <= 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) Ro[k] = 0.f;
    to[0] = to[1] = to[2] = 0.f;
    return;
  }
  double m[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) m[k] = 0.0;
  const int nc = proc_chunks_of(n);
  for (int c = 0; c < nc; ++c) {
    const double* pr = partials + ((size_t)part * RAP_PROC_CHUNKS + c) * 16;
#pragma unroll
    for (int k = 0; k < 15; ++k) m[k] += pr[k];
  }
  rap_kabsch_from_moments(m, n, Ro, to);
}

__global__ __launch_bounds__(256) void rigid_apply_kernel(const float* __restrict__ src, const float* __restrict__ Rm,
                                                          const float* __restrict__ tv, const int32_t* __restrict__ off,
                                                          float* __restrict__ out, const float* __restrict__ x1, float w0,
                                                          float w1, float* __restrict__ traj, int blend) {
  const int part = blockIdx.y;
  const int a = off[part], n = off[part + 1] - a;
  if (n <= 0) return;
  const float* R = Rm + (size_t)part * 9;
  const float r00 = R[0], r01 = R[1], r02 = R[2], r10 = R[3], r11 = R[4], r12 = R[5], r20 = R[6], r21 = R[7], r22 = R[8];
  const float t0 = tv[part * 3 + 0], t1 = tv[part * 3 + 1], t2 = tv[part * 3 + 2];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const size_t p = (size_t)(a + i) * 3;
    const float s0 = src[p], s1 = src[p + 1], s2 = src[p + 2];
    // x R^T + t : matmul (fma chain over k) then the translation add, as `parts_source @ rot.t() + trans`
    float y0 = fmaf(s2, r02, fmaf(s1, r01, s0 * r00));
    float y1 = fmaf(s2, r12, fmaf(s1, r11, s0 * r10));
    float y2 = fmaf(s2, r22, fmaf(s1, r21, s0 * r20));
    asm volatile("" : "+v"(y0), "+v"(y1), "+v"(y2));   // the matmul result is rounded before the translation add
    y0 += t0; y1 += t1; y2 += t2;
    if (blend) {
      // x_t = x0_rigid * (1 - t + dt) + x_1 * (t - dt)   (sampler.py:60), separate roundings
      y0 = mul_rn_nofuse(y0, w0) + mul_rn_nofuse(x1[p], w1);
      y1 = mul_rn_nofuse(y1, w0) + mul_rn_nofuse(x1[p + 1], w1);
      y2 = mul_rn_nofuse(y2, w0) + mul_rn_nofuse(x1[p + 2], w1);
    }
    out[p] = y0; out[p + 1] = y1; out[p + 2] = y2;
    if (traj) { traj[p] = y0; traj[p + 1] = y1; traj[p + 2] = y2; }
  }
}

int launch_procrustes_fit(hipStream_t stream, const float* src, const float* tgt, const int32_t* part_offsets, int nparts,
                          float* R_out, float* t_out, double* partials) {
  if (nparts <= 0) return RAP_OK;
  hipLaunchKernelGGL(procrustes_moments_kernel, dim3(RAP_PROC_CHUNKS, nparts), dim3(256), 0, stream, src, tgt, part_offsets,
                     partials);
  RAP_LAUNCH_CHECK();
  hipLaunchKernelGGL(procrustes_solve_kernel, dim3((nparts + 63) / 64), dim3(64), 0, stream, partials, part_offsets, nparts,
                     R_out, t_out);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

int launch_rigid_apply(hipStream_t stream, const float* src, const float* R, const float* t, const int32_t* part_offsets,
                       int nparts, float* out, const float* x1, float w0, float w1, float* traj_slot_or_null, int blend) {
  if (nparts <= 0) return RAP_OK;
  hipLaunchKernelGGL(rigid_apply_kernel, dim3(RAP_PROC_CHUNKS, nparts), dim3(256), 0, stream, src, R, t, part_offsets, out, x1,
                     w0, w1, traj_slot_or_null, blend);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
