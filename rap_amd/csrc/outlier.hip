// Statistical outlier removal on device (preprocessing in front of FPS / MiniSpinNet): replaces Open3D's
// PointCloud.remove_statistical_outlier(nb_neighbors, std_ratio) as dataset_process/extract_sample_features.py:378-385 calls it
// (nb_neighbors = 20, std_ratio = 2.5).  Open3D (0.18, pinned by the reference's environment) is not in the mount: the rule below
// restates its published algorithm -- PARITY UNPINNED for this function:
//   d_i      = mean distance from point i to its nb_neighbors nearest points, the point itself included (KNN of the point in its own
//              cloud returns it first, at distance 0)
//   mean     = sum_{d_i > 0} d_i / N,   std = sqrt( sum_{d_i > 0} (d_i - mean)^2 / (N - 1) )        (Bessel's correction)
//   inlier_i = d_i > 0  and  d_i < mean + std_ratio * std;   the indices of the inliers are returned in ascending order.
// Kernel 1 is the hot one: brute-force k nearest neighbours, one query per lane, the cloud streamed through the LDS in tiles of
// 1024 points (float4, 16 KB), each lane keeping its K smallest squared distances as a sorted register array that is updated with
// a branch-free min/max ladder only when a candidate beats the current K-th (rare after the first tiles).  N^2 direct-difference
// fp32 distances, no N x N matrix in HBM: MFMA is useless here (3-d points), the bound is VALU issue (8 ops per pair).
#include "kernels.h"

#define OL_TILE 1024
template <int K>
__global__ __launch_bounds__(256) void knn_mean_dist_kernel(const float* __restrict__ pts, int N, int k, double* __restrict__ mean_dist) {
  __shared__ float4 tile[OL_TILE];
  const int q = blockIdx.x * 256 + threadIdx.x;
  const bool live = q < N;
  const int qi = live ? q : N - 1;
  const float qx = pts[3 * (size_t)qi], qy = pts[3 * (size_t)qi + 1], qz = pts[3 * (size_t)qi + 2];
  float best[K];
#pragma unroll
  for (int j = 0; j < K; ++j) best[j] = __builtin_inff();
  for (int t0 = 0; t0 < N; t0 += OL_TILE) {
    __syncthreads();
    for (int j = threadIdx.x; j < OL_TILE; j += 256) {
      const int i = t0 + j;
      tile[j] = i < N ? make_float4(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], 0.f)
                      : make_float4(0.f, 0.f, 0.f, __builtin_inff());
    }
    __syncthreads();
    const int cnt = N - t0 < OL_TILE ? N - t0 : OL_TILE;
    for (int j = 0; j < cnt; ++j) {
      const float4 p = tile[j];                                  // same address for every lane: LDS broadcast
      const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
      const float d2 = dx * dx + dy * dy + dz * dz;
      if (d2 < best[K - 1]) {
#pragma unroll
        for (int m = K - 1; m > 0; --m) best[m] = fminf(fmaxf(d2, best[m - 1]), best[m]);
        best[0] = fminf(d2, best[0]);
      }
    }
  }
  if (!live) return;
  double s = 0.0; int n = 0;
#pragma unroll
  for (int j = 0; j < K; ++j)
    if (j < k && best[j] < __builtin_inff()) { s += (double)sqrtf(best[j]); ++n; }
  mean_dist[q] = n > 0 ? s / (double)n : -1.0;
}

// one block: stats[0] = mean, stats[1] = std (Bessel), stats[2] = threshold
__global__ __launch_bounds__(1024) void outlier_stats_kernel(const double* __restrict__ d, int N, double std_ratio, double* __restrict__ stats) {
  __shared__ double sh[16];
  __shared__ double s_mean;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s = 0.0;
  for (int i = threadIdx.x; i < N; i += 1024) { const double v = d[i]; s += v > 0.0 ? v : 0.0; }
  s = wave_sum_d(s);
  if (lane == 0) sh[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < 16; ++w) t += sh[w]; s_mean = t / (double)N; }
  __syncthreads();
  const double mean = s_mean;
  double q = 0.0;
  for (int i = threadIdx.x; i < N; i += 1024) { const double v = d[i]; q += v > 0.0 ? (v - mean) * (v - mean) : 0.0; }
  q = wave_sum_d(q);
  __syncthreads();
  if (lane == 0) sh[wave] = q;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0; for (int w = 0; w < 16; ++w) t += sh[w];
    const double sd = N > 1 ? sqrt(t / (double)(N - 1)) : 0.0;
    stats[0] = mean; stats[1] = sd; stats[2] = mean + std_ratio * sd;
  }
}

// ordered compaction of the inlier predicate: per-block counts -> one-block exclusive scan -> emit (ascending indices)
#define OL_CHUNK 4096
__device__ __forceinline__ bool ol_inlier(const double* d, int i, double thr) { const double v = d[i]; return v > 0.0 && v < thr; }
__global__ __launch_bounds__(256) void outlier_count_kernel(const double* __restrict__ d, int N, const double* __restrict__ stats,
                                                            unsigned int* __restrict__ block_cnt) {
  __shared__ unsigned int red[4];
  const double thr = stats[2];
  unsigned int c = 0;
  for (int j = threadIdx.x; j < OL_CHUNK; j += 256) { const int i = blockIdx.x * OL_CHUNK + j; c += (i < N && ol_inlier(d, i, thr)) ? 1u : 0u; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_cnt[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(1024) void outlier_scan_kernel(unsigned int* __restrict__ block_cnt, int nblk, int32_t* __restrict__ total) {
  __shared__ unsigned int wsum[16];
  __shared__ unsigned int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b0 = 0; b0 < nblk; b0 += 1024) {
    const int b = b0 + threadIdx.x;
    const unsigned int x = b < nblk ? block_cnt[b] : 0u;
    unsigned int inc = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned int y = __shfl_up(inc, o, 64); if (lane >= o) inc += y; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const unsigned int carry = carry_s;
    if (b < nblk) block_cnt[b] = carry + woff + inc - x;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) total[0] = (int32_t)carry_s;
}
__global__ __launch_bounds__(256) void outlier_emit_kernel(const double* __restrict__ d, int N, const double* __restrict__ stats,
                                                           const unsigned int* __restrict__ block_off, int64_t* __restrict__ idx_out) {
  __shared__ unsigned int wcnt[4];
  __shared__ unsigned int run_s;
  const double thr = stats[2];
  if (threadIdx.x == 0) run_s = block_off[blockIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j0 = 0; j0 < OL_CHUNK; j0 += 256) {
    const int i = blockIdx.x * OL_CHUNK + j0 + threadIdx.x;
    const bool in = i < N && ol_inlier(d, i, thr);
    const unsigned long long m = __ballot(in);
    if (lane == 0) wcnt[wave] = __popcll(m);
    __syncthreads();
    unsigned int before = run_s;
    for (int w = 0; w < wave; ++w) before += wcnt[w];
    if (in) idx_out[before + __popcll(m & ((1ull << lane) - 1ull))] = (int64_t)i;
    __syncthreads();
    if (threadIdx.x == 0) run_s += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
}

size_t outlier_workspace_bytes(long N) {
  const long nblk = (N + OL_CHUNK - 1) / OL_CHUNK;
  return align_up((size_t)N * 8, 256) + 256 + align_up((size_t)(nblk + 1) * 4, 256);
}

int launch_statistical_outliers(hipStream_t stream, const float* pts, long N, int nb_neighbors, double std_ratio, int64_t* idx_out,
                                int32_t* count_out, double* stats_out, void* ws) {
  char* w = (char*)ws;
  double* md = (double*)w; w += align_up((size_t)N * 8, 256);
  double* stats = (double*)w; w += 256;
  unsigned int* block_cnt = (unsigned int*)w;
  const int n = (int)N;
  const unsigned grid = (unsigned)((N + 255) / 256);
  if (nb_neighbors <= 20) hipLaunchKernelGGL(knn_mean_dist_kernel<20>, dim3(grid), dim3(256), 0, stream, pts, n, nb_neighbors, md);
  else hipLaunchKernelGGL(knn_mean_dist_kernel<32>, dim3(grid), dim3(256), 0, stream, pts, n, nb_neighbors, md);
  RAP_LAUNCH_CHECK();
  hipLaunchKernelGGL(outlier_stats_kernel, dim3(1), dim3(1024), 0, stream, md, n, std_ratio, stats);
  RAP_LAUNCH_CHECK();
  const int nblk = (int)((N + OL_CHUNK - 1) / OL_CHUNK);
  hipLaunchKernelGGL(outlier_count_kernel, dim3(nblk), dim3(256), 0, stream, md, n, stats, block_cnt);
  RAP_LAUNCH_CHECK();
  hipLaunchKernelGGL(outlier_scan_kernel, dim3(1), dim3(1024), 0, stream, block_cnt, nblk, count_out);
  RAP_LAUNCH_CHECK();
  hipLaunchKernelGGL(outlier_emit_kernel, dim3(nblk), dim3(256), 0, stream, md, n, stats, block_cnt, idx_out);
  RAP_LAUNCH_CHECK();
  if (stats_out) RAP_HIP_CHECK(hipMemcpyAsync(stats_out, stats, 3 * sizeof(double), hipMemcpyDeviceToDevice, stream));
  return RAP_OK;
}
