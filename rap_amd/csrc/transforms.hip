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
