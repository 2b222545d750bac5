// Generation selection by rigidity on device: replaces the triple Python loop of the
// reference's test_step (rectified_point_flow/modeling.py:456-592) and compute_rigidity_rmse
// (rectified_point_flow/eval/metrics.py:511-622), which run `steps x generations x B x P` small torch ops with a host
// sync each.
//
//   rigidity RMSE of one sample (metrics.py:569-616):  e_i = || x_i R_p^T + t_p - y_i ||^2 over the points of every
//   non-empty part p;  default: sqrt(mean over all points);  average_per_part: mean over parts of sqrt(mean over the
//   part);  inf if the sample has no points;  times scales[b] when scales are given (metres).
//   use_average_rigidity_rmse (modeling.py:466-500): the mean over all trajectory steps of that RMSE, each step with
//   its own Procrustes fit cond -> x0_hat(step).
//   selection (modeling.py:518, 560-592): per sample argmin over generations; gather that generation's final cloud,
//   rotations and translations.
//
// Kernels (HBM-bound, 24 B read per point and step; no atomics, no host sync):
//   rigid_sqerr: grid (16 chunks, parts) -> one fp64 partial per (part, chunk); per-point error in fp32 exactly as
//                the reference forms it (fp32 matmul, add t, subtract, square, sum of three);
//   rigidity_finish: one lane per sample folds the partials of its parts in fixed order;
//   step_mean / argmin / gather: trivially small.
#include "kernels.h"

__global__ __launch_bounds__(256) void rigid_sqerr_kernel(const float* __restrict__ src, const float* __restrict__ tgt,
                                                          const float* __restrict__ Rm, const float* __restrict__ tv,
                                                          const int32_t* __restrict__ off, double* __restrict__ partials) {
  __shared__ double red[4];
  const int part = blockIdx.y, chunk = blockIdx.x;
  const int a = off[part], n = off[part + 1] - a;
  const long lo = a + (long)n * chunk / RAP_PROC_CHUNKS;
  const long hi = a + (long)n * (chunk + 1) / RAP_PROC_CHUNKS;
  const float* R = Rm + (size_t)part * 9;
  const float r00 = R[0], r01 = R[1], r02 = R[2], r10 = R[3], r11 = R[4], r12 = R[5], r20 = R[6], r21 = R[7], r22 = R[8];
  const float t0 = tv[part * 3 + 0], t1 = tv[part * 3 + 1], t2 = tv[part * 3 + 2];
  double acc = 0.0;
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const float s0 = src[i * 3 + 0], s1 = src[i * 3 + 1], s2 = src[i * 3 + 2];
    float y0 = fmaf(s2, r02, fmaf(s1, r01, s0 * r00));
    float y1 = fmaf(s2, r12, fmaf(s1, r11, s0 * r10));
    float y2 = fmaf(s2, r22, fmaf(s1, r21, s0 * r20));
    asm volatile("" : "+v"(y0), "+v"(y1), "+v"(y2));   // matmul result rounded before the translation add (as rigid_apply)
    const float d0 = (y0 + t0) - tgt[i * 3 + 0], d1 = (y1 + t1) - tgt[i * 3 + 1], d2 = (y2 + t2) - tgt[i * 3 + 2];
    acc += (double)(d0 * d0 + d1 * d1 + d2 * d2);
  }
  const double v = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) partials[(size_t)part * RAP_PROC_CHUNKS + chunk] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void rigidity_finish_kernel(const double* __restrict__ partials, const int32_t* __restrict__ off,
                                                             int B, int P, const float* __restrict__ scales, int average_per_part,
                                                             float* __restrict__ out) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  double total = 0.0, rmse_sum = 0.0;
  long npts = 0;
  int nparts = 0;
  for (int p = 0; p < P; ++p) {
    const int part = b * P + p;
    const int n = off[part + 1] - off[part];
    if (n <= 0) continue;                       // metrics.py:566-567, 594-595: empty parts are skipped
    double s = 0.0;
    for (int c = 0; c < RAP_PROC_CHUNKS; ++c) s += partials[(size_t)part * RAP_PROC_CHUNKS + c];
    total += s; npts += n; ++nparts;
    rmse_sum += sqrt(s / (double)n);
  }
  float r;
  if (nparts == 0) r = __builtin_inff();        // metrics.py:589, 616
  else if (average_per_part) r = (float)(rmse_sum / (double)nparts);
  else r = (float)sqrt(total / (double)npts);
  if (scales) r *= scales[b];                   // metrics.py:619-620
  out[b] = r;
}

// out[b] = mean over steps of per_step[s][b]   (torch.stack(step_rmses).mean(dim=0), modeling.py:489)
__global__ __launch_bounds__(64) void step_mean_kernel(const float* __restrict__ per_step, int S, int B, float* __restrict__ out) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  double s = 0.0;
  for (int k = 0; k < S; ++k) s += (double)per_step[(size_t)k * B + b];
  out[b] = (float)(s / (double)S);
}

// best[b] = first index of the minimum (torch.argmin, modeling.py:518: rigidity) or of the maximum (torch.argmax, :601: overlap
// ratio) over generations
__global__ __launch_bounds__(64) void argmin_generation_kernel(const float* __restrict__ rmse, int G, int B, int largest,
                                                               int32_t* __restrict__ best) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  int bi = 0;
  float bv = rmse[b];
  bool bn = bv != bv;                      // torch.argmin / argmax treat NaN as the extremum: the FIRST NaN wins
  for (int g = 1; g < G && !bn; ++g) {
    const float v = rmse[(size_t)g * B + b];
    const bool vn = v != v;
    if (vn || (largest ? (v > bv) : (v < bv))) { bv = v; bi = g; bn = vn; }
  }
  best[b] = bi;
}

// cloud_out[token] = clouds[best[sample of token]][token];  R_out[b] = R[best[b]][b];  t_out[b] = t[best[b]][b]
__global__ __launch_bounds__(256) void gather_generation_kernel(const float* __restrict__ clouds, const float* __restrict__ R,
                                                                const float* __restrict__ t, const int32_t* __restrict__ best,
                                                                const int32_t* __restrict__ cu_batch, int B, int P, long TP,
                                                                float* __restrict__ cloud_out, float* __restrict__ R_out,
                                                                float* __restrict__ t_out) {
  const int b = blockIdx.y;
  const int g = best[b];
  const long a = cu_batch[b], e = cu_batch[b + 1];
  for (long i = a * 3 + blockIdx.x * 256 + threadIdx.x; i < e * 3; i += (long)gridDim.x * 256)
    cloud_out[i] = clouds[(size_t)g * TP * 3 + i];
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < P * 9; i += 256) R_out[(size_t)b * P * 9 + i] = R[((size_t)g * B + b) * P * 9 + i];
    for (int i = threadIdx.x; i < P * 3; i += 256) t_out[(size_t)b * P * 3 + i] = t[((size_t)g * B + b) * P * 3 + i];
  }
}

int launch_rigidity_rmse(hipStream_t stream, const float* src, const float* tgt, const float* R, const float* t,
                         const int32_t* part_offsets, int B, int P, const float* scales, int average_per_part, float* out,
                         double* partials) {
  if (B <= 0 || P <= 0) return RAP_OK;
  hipLaunchKernelGGL(rigid_sqerr_kernel, dim3(RAP_PROC_CHUNKS, B * P), dim3(256), 0, stream, src, tgt, R, t, part_offsets, partials);
  RAP_LAUNCH_CHECK();
  hipLaunchKernelGGL(rigidity_finish_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, partials, part_offsets, B, P, scales,
                     average_per_part, out);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
int launch_step_mean(hipStream_t stream, const float* per_step, int S, int B, float* out) {
  hipLaunchKernelGGL(step_mean_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, per_step, S, B, out);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
int launch_select_generation(hipStream_t stream, const float* rmse, int G, int B, int P, long TP, const int32_t* cu_batch,
                             const float* clouds, const float* R, const float* t, int largest, int32_t* best, float* cloud_out,
                             float* R_out, float* t_out) {
  hipLaunchKernelGGL(argmin_generation_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, rmse, G, B, largest, best);
  RAP_LAUNCH_CHECK();
  if (clouds && cloud_out) {
    hipLaunchKernelGGL(gather_generation_kernel, dim3(16, B), dim3(256), 0, stream, clouds, R, t, best, cu_batch, B, P, TP,
                       cloud_out, R_out, t_out);
    RAP_LAUNCH_CHECK();
  }
  return RAP_OK;
}
