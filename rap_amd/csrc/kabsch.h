// 3x3 Kabsch solve from raw moments, fp64, shared by the device solve kernel (procrustes.hip) and a
// host unit test (tests/host/kabsch_host.cpp compiles this header with g++ so the Jacobi SVD can be
// checked against LAPACK without a GPU).  Plain C++; the only HIP-specific token is the qualifier.
//
// Reference: solve_procrustes (rectified_point_flow/procrustes.py:6-37):
//   H = (src - mu_s)^T (tgt - mu_t);  U S V^T = svd(H);  R = V U^T;
//   if det R < 0: V^T[-1,:] *= -1 (the smallest singular value's row);  t = mu_t - mu_s R^T.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define RAP_HD __host__ __device__
#else
#define RAP_HD
#endif

RAP_HD inline double rap_det3(const double M[3][3]) {
  return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
         M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}

RAP_HD inline void rap_swap_cols(double A[3][3], double V[3][3], double sig[3], int p, int q) {
  for (int i = 0; i < 3; ++i) {
    double t = A[i][p]; A[i][p] = A[i][q]; A[i][q] = t;
    t = V[i][p]; V[i][p] = V[i][q]; V[i][q] = t;
  }
  double t = sig[p]; sig[p] = sig[q]; sig[q] = t;
}

// One-sided (Hestenes) Jacobi rotation of columns p,q of A (and V).  Returns the relative
// off-diagonal size |a_p . a_q| / (|a_p||a_q|) before the rotation (0 if skipped).
RAP_HD inline double rap_jacobi_pair(double A[3][3], double V[3][3], int p, int q) {
  double alpha = 0, beta = 0, gamma = 0;
  for (int i = 0; i < 3; ++i) { alpha += A[i][p] * A[i][p]; beta += A[i][q] * A[i][q]; gamma += A[i][p] * A[i][q]; }
  const double nrm = sqrt(alpha * beta);
  if (!(fabs(gamma) > 1e-16 * nrm) || fabs(gamma) < 1e-300) return 0.0;
  const double zeta = (beta - alpha) / (2.0 * gamma);
  const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
  for (int i = 0; i < 3; ++i) {
    const double ap = A[i][p], aq = A[i][q];
    A[i][p] = c * ap - s * aq; A[i][q] = s * ap + c * aq;
    const double vp = V[i][p], vq = V[i][q];
    V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
  }
  return fabs(gamma) / nrm;
}

// R (row-major 3x3) = V diag(1,1,sign) U^T for H = U S V^T, sign chosen so det R = +1.
RAP_HD inline void rap_kabsch_from_H(const double Hm[3][3], double R[3][3]) {
  double A[3][3], V[3][3], sig[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { A[i][j] = Hm[i][j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = rap_jacobi_pair(A, V, 0, 1);
    off = fmax(off, rap_jacobi_pair(A, V, 0, 2));
    off = fmax(off, rap_jacobi_pair(A, V, 1, 2));
    if (off < 1e-15) break;
  }
  for (int j = 0; j < 3; ++j) sig[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
  if (sig[0] < sig[1]) rap_swap_cols(A, V, sig, 0, 1);
  if (sig[0] < sig[2]) rap_swap_cols(A, V, sig, 0, 2);
  if (sig[1] < sig[2]) rap_swap_cols(A, V, sig, 1, 2);
  const double tiny = 1e-12 * fmax(sig[0], 1e-300);
  if (sig[0] > 1e-300) {
    for (int i = 0; i < 3; ++i) A[i][0] /= sig[0];
  } else {
    A[0][0] = 1; A[1][0] = 0; A[2][0] = 0;
  }
  if (sig[1] > tiny) {
    for (int i = 0; i < 3; ++i) A[i][1] /= sig[1];
  } else {  // rank <= 1: any unit vector orthogonal to u0
    const double ax = fabs(A[0][0]), ay = fabs(A[1][0]), az = fabs(A[2][0]);
    double e0 = 0, e1 = 0, e2 = 0;
    if (ax <= ay && ax <= az) e0 = 1; else if (ay <= az) e1 = 1; else e2 = 1;
    const double dp = e0 * A[0][0] + e1 * A[1][0] + e2 * A[2][0];
    const double w0 = e0 - dp * A[0][0], w1 = e1 - dp * A[1][0], w2 = e2 - dp * A[2][0];
    const double nw = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    A[0][1] = w0 / nw; A[1][1] = w1 / nw; A[2][1] = w2 / nw;
  }
  if (sig[2] > tiny) {
    for (int i = 0; i < 3; ++i) A[i][2] /= sig[2];
  } else {  // rank <= 2: complete the basis (its sign is fixed by the determinant rule below)
    A[0][2] = A[1][0] * A[2][1] - A[2][0] * A[1][1];
    A[1][2] = A[2][0] * A[0][1] - A[0][0] * A[2][1];
    A[2][2] = A[0][0] * A[1][1] - A[1][0] * A[0][1];
  }
  // det(V U^T) = det V * det U; flip the smallest-sigma pair when negative (procrustes.py:31-33)
  const double dsign = (rap_det3(V) * rap_det3(A) < 0.0) ? -1.0 : 1.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i][j] = V[i][0] * A[j][0] + V[i][1] * A[j][1] + dsign * V[i][2] * A[j][2];
}

// m[0:3] = sum s, m[3:6] = sum t, m[6:15] = sum s_i t_j (row-major), n points.
RAP_HD inline void rap_kabsch_from_moments(const double m[15], int n, float R_out[9], float t_out[3]) {
  const double inv = 1.0 / (double)n;
  const double ms[3] = {m[0] * inv, m[1] * inv, m[2] * inv};
  const double mt[3] = {m[3] * inv, m[4] * inv, m[5] * inv};
  double Hm[3][3], R[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Hm[i][j] = m[6 + 3 * i + j] - (double)n * ms[i] * mt[j];
  rap_kabsch_from_H(Hm, R);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) R_out[3 * i + j] = (float)R[i][j];
    // t = mu_t - mu_s R^T (row vectors)  ->  t_i = mt_i - sum_j R[i][j] ms_j
    t_out[i] = (float)(mt[i] - (R[i][0] * ms[0] + R[i][1] * ms[1] + R[i][2] * ms[2]));
  }
}
