// Output side of the path: the per-part 4x4 transforms the reference writes to
// `<dataset>_sample<idx>_<generation>_part<pid>_transform.txt` (rectified_point_flow/eval/evaluator.py:383-490), consumed by
// demo.py:1332-1342.  Predicted pose relative to the ground-truth pose, in metres, optionally taken back out of the global
// normalisation frame:
//     R_rel = R_pred R_gt^T ;  t_rel = s t_pred - (s t_gt) R_rel^T ;  M = [R_rel | t_rel ; 0 0 0 1]   (evaluator.py:444-461)
//     M <- M * inv([R_global | t_global ; 0 0 0 1])                                                    (:464-474)
// One lane per (sample, part); fp64 like the reference's numpy code, the 4x4 is rounded to fp32 where the reference stores it
// in a float32 matrix.  Parts with no points get an all-zero block (the reference writes no file for them).
#include "kernels.h"

__global__ __launch_bounds__(64) void relative_transform_kernel(const float* __restrict__ R_pred, const float* __restrict__ t_pred,
                                                                const float* __restrict__ R_gt, const float* __restrict__ t_gt,
                                                                const float* __restrict__ scales, const int64_t* __restrict__ ppp,
                                                                int B, int P, const float* __restrict__ R_glob,
                                                                const float* __restrict__ t_glob, float* __restrict__ out) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= B * P) return;
  const int b = i / P;
  float* M = out + (size_t)i * 16;
  if (ppp[i] <= 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) M[k] = 0.f;
    return;
  }
  const float* Rp = R_pred + (size_t)i * 9; const float* Rg = R_gt + (size_t)i * 9;
  const double s = (double)scales[b];
  double tp[3], tg[3], RrT[3][3];     // RrT = R_rel^T = R_gt R_pred^T
#pragma unroll
  for (int k = 0; k < 3; ++k) { tp[k] = (double)t_pred[(size_t)i * 3 + k] * s; tg[k] = (double)t_gt[(size_t)i * 3 + k] * s; }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) a += (double)Rg[r * 3 + k] * (double)Rp[c * 3 + k];
      RrT[r][c] = a;
    }
  float m[4][4];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) m[r][c] = (float)RrT[c][r];                                   // R_rel
    m[r][3] = (float)(tp[r] - (tg[0] * RrT[0][r] + tg[1] * RrT[1][r] + tg[2] * RrT[2][r]));   // t_pred_m - t_gt_m @ R_rel^T
  }
  m[3][0] = m[3][1] = m[3][2] = 0.f; m[3][3] = 1.f;
  if (R_glob && t_glob) {
    // inv([G | g ; 0 1]) = [G^-1 | -G^-1 g ; 0 1]; G^-1 by the adjugate (the reference calls the general np.linalg.inv)
    double G[3][3], Gi[3][3], g[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      g[r] = (double)t_glob[(size_t)b * 3 + r];
#pragma unroll
      for (int c = 0; c < 3; ++c) G[r][c] = (double)R_glob[(size_t)b * 9 + r * 3 + c];
    }
    const double det = G[0][0] * (G[1][1] * G[2][2] - G[1][2] * G[2][1]) - G[0][1] * (G[1][0] * G[2][2] - G[1][2] * G[2][0]) +
                       G[0][2] * (G[1][0] * G[2][1] - G[1][1] * G[2][0]);
    const double id = 1.0 / det;
    Gi[0][0] = (G[1][1] * G[2][2] - G[1][2] * G[2][1]) * id; Gi[0][1] = (G[0][2] * G[2][1] - G[0][1] * G[2][2]) * id;
    Gi[0][2] = (G[0][1] * G[1][2] - G[0][2] * G[1][1]) * id; Gi[1][0] = (G[1][2] * G[2][0] - G[1][0] * G[2][2]) * id;
    Gi[1][1] = (G[0][0] * G[2][2] - G[0][2] * G[2][0]) * id; Gi[1][2] = (G[0][2] * G[1][0] - G[0][0] * G[1][2]) * id;
    Gi[2][0] = (G[1][0] * G[2][1] - G[1][1] * G[2][0]) * id; Gi[2][1] = (G[0][1] * G[2][0] - G[0][0] * G[2][1]) * id;
    Gi[2][2] = (G[0][0] * G[1][1] - G[0][1] * G[1][0]) * id;
    double gi[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) gi[r] = -(Gi[r][0] * g[0] + Gi[r][1] * g[1] + Gi[r][2] * g[2]);
    float o[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) o[r][c] = (float)((double)m[r][0] * Gi[0][c] + (double)m[r][1] * Gi[1][c] + (double)m[r][2] * Gi[2][c]);
      o[r][3] = (float)((double)m[r][0] * gi[0] + (double)m[r][1] * gi[1] + (double)m[r][2] * gi[2] + (double)m[r][3]);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) m[r][c] = o[r][c];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) M[r * 4 + c] = m[r][c];
}

int launch_relative_transforms(hipStream_t stream, const float* R_pred, const float* t_pred, const float* R_gt, const float* t_gt,
                               const float* scales, const int64_t* ppp, int B, int P, const float* R_glob, const float* t_glob,
                               float* out) {
  if (B <= 0 || P <= 0) return RAP_OK;
  hipLaunchKernelGGL(relative_transform_kernel, dim3((B * P + 63) / 64), dim3(64), 0, stream, R_pred, t_pred, R_gt, t_gt, scales, ppp,
                     B, P, R_glob, t_glob, out);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

// ---------------------------------------------------------------------------------------------
// compute_transform_errors, no-ICP branch (round 6; reference eval/metrics.py:165-303, the RRE / RTE BASELINE.json's "SE(3) err" is defined
// by; the ICP branch is "for the interchangable case, which does not apply for point cloud registration tasks", metrics.py:264).
// Per sample: the (first) anchor part's ground-truth and predicted poses are inverted, every non-anchor, non-empty part's poses are taken
// relative to them, delta_R = R_gt_rel^T R_pred_rel, delta_t = (t_pred_rel - t_gt_rel) scale;
//   RE = deg(acos(clamp((tr delta_R - 1) / 2, -1, 1))),  TE = |delta_t|;  means over the valid parts (0 / 0 = NaN as in the reference).
// One block per sample, one lane per part; fp32 products in the reference's association (3 x 3 matmuls as fma chains over k), the
// per-sample sums in part order by one lane (deterministic).  matched (B,P) int64 or null re-orders the PREDICTED poses (:223-227).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void te_mat3_mul(const float* A, const float* B, float* C) {       // C = A B, row-major
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[r * 3 + c] = fmaf(A[r * 3 + 2], B[6 + c], fmaf(A[r * 3 + 1], B[3 + c], A[r * 3] * B[c]));
}
__device__ __forceinline__ void te_mat3_vec(const float* A, const float* v, float* o) {
#pragma unroll
  for (int r = 0; r < 3; ++r) o[r] = fmaf(A[r * 3 + 2], v[2], fmaf(A[r * 3 + 1], v[1], A[r * 3] * v[0]));
}
__device__ __forceinline__ void te_transpose(const float* A, float* T) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) T[r * 3 + c] = A[c * 3 + r];
}

__global__ __launch_bounds__(64) void transform_errors_kernel(const float* __restrict__ R_gt, const float* __restrict__ t_gt,
                                                              const float* __restrict__ R_pred, const float* __restrict__ t_pred,
                                                              const int64_t* __restrict__ ppp, const uint8_t* __restrict__ anchor,
                                                              const int64_t* __restrict__ matched, const float* __restrict__ scale, int P,
                                                              float* __restrict__ rot_pp, float* __restrict__ trans_pp,
                                                              float* __restrict__ rot_mean, float* __restrict__ trans_mean) {
  const int b = blockIdx.x;
  const size_t row = (size_t)b * P;
  auto pred_index = [&](int p) -> size_t {
    long q = matched ? (long)matched[row + p] : (long)p;
    q = q < 0 ? 0 : q >= P ? P - 1 : q;              // an out-of-range match cannot index outside the sample's rows
    return row + (size_t)q;
  };
  // the first anchor part (metrics.py:239-242), found by every lane the same way
  int a = -1;
  for (int p = 0; p < P; ++p)
    if (anchor[row + p]) { a = p; break; }
  float Rag_inv[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}, tag_inv[3] = {0.f, 0.f, 0.f};
  float Rap_inv[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}, tap_inv[3] = {0.f, 0.f, 0.f};
  if (a >= 0) {
    float tmp[3];
    te_transpose(R_gt + (row + a) * 9, Rag_inv);
    te_mat3_vec(Rag_inv, t_gt + (row + a) * 3, tmp);
    tag_inv[0] = -tmp[0]; tag_inv[1] = -tmp[1]; tag_inv[2] = -tmp[2];          // (-R^T) @ t evaluates the product of the negated matrix: same magnitude
    const size_t ia = pred_index(a);
    te_transpose(R_pred + ia * 9, Rap_inv);
    te_mat3_vec(Rap_inv, t_pred + ia * 3, tmp);
    tap_inv[0] = -tmp[0]; tap_inv[1] = -tmp[1]; tap_inv[2] = -tmp[2];
  }
  const float s = scale ? scale[b] : 1.0f;
  for (int p = threadIdx.x; p < P; p += 64) {
    float re = 0.f, te = 0.f;
    if (ppp[row + p] != 0 && !anchor[row + p]) {
      const size_t ip = pred_index(p);
      float Rg_rel[9], Rp_rel[9], tg_rel[3], tp_rel[3], RgT[9], dR[9];
      te_mat3_mul(Rag_inv, R_gt + (row + p) * 9, Rg_rel);
      te_mat3_vec(Rag_inv, t_gt + (row + p) * 3, tg_rel);
      te_mat3_mul(Rap_inv, R_pred + ip * 9, Rp_rel);
      te_mat3_vec(Rap_inv, t_pred + ip * 3, tp_rel);
#pragma unroll
      for (int k = 0; k < 3; ++k) { tg_rel[k] += tag_inv[k]; tp_rel[k] += tap_inv[k]; }
      te_transpose(Rg_rel, RgT);
      te_mat3_mul(RgT, Rp_rel, dR);
      const float tr = (dR[0] + dR[4]) + dR[8];
      float c = 0.5f * (tr - 1.0f);
      c = c < -1.0f ? -1.0f : c > 1.0f ? 1.0f : c;
      re = acosf(c) * 57.29577951308232f;
      const float d0 = (tp_rel[0] - tg_rel[0]) * s, d1 = (tp_rel[1] - tg_rel[1]) * s, d2 = (tp_rel[2] - tg_rel[2]) * s;
      te = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    }
    rot_pp[row + p] = re; trans_pp[row + p] = te;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float sr = 0.f, st = 0.f; int n = 0;
    for (int p = 0; p < P; ++p) {
      sr += rot_pp[row + p]; st += trans_pp[row + p];
      n += (ppp[row + p] != 0 && !anchor[row + p]) ? 1 : 0;
    }
    rot_mean[b] = sr / (float)n;             // 0 / 0 = NaN for a sample without a movable part, as the reference's division
    trans_mean[b] = st / (float)n;
  }
}

int launch_transform_errors(hipStream_t stream, const float* R_gt, const float* t_gt, const float* R_pred, const float* t_pred,
                            const int64_t* ppp, const uint8_t* anchor, const int64_t* matched, const float* scale, int B, int P,
                            float* rot_pp, float* trans_pp, float* rot_mean, float* trans_mean) {
  if (B <= 0 || P <= 0) return RAP_OK;
  hipLaunchKernelGGL(transform_errors_kernel, dim3(B), dim3(64), 0, stream, R_gt, t_gt, R_pred, t_pred, ppp, anchor, matched, scale, P,
                     rot_pp, trans_pp, rot_mean, trans_mean);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
