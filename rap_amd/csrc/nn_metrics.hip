// Nearest-neighbour registration metrics on device: chamfer RMSE per object and correspondence RMSE
// of a scan pair -- compute_cd (rectified_point_flow/eval/metrics.py:14-48, a pytorch3d chamfer_distance per object in a Python
// loop) and compute_correspondence_rmse (:386-469, a dense N_s x N_t torch.cdist matrix + row minima).
//
// One kernel does the N^2 work for both: nn_query_kernel -- a block owns 256 query points (one per lane, in registers) and streams
// the candidate set through LDS in tiles of 256 float4 (one ds_read_b128 per candidate, broadcast to the wave); direct-difference
// fp32 squared distances, running minimum and FIRST arg-minimum (torch.min tie rule); nothing N^2 ever reaches HBM.
#include "kernels.h"

#define NN_TILE 256

__global__ void nn_worklist_kernel(const int32_t* __restrict__ cu, int B, int swap_unused, NnWork* __restrict__ items, int max_items) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int n = 0;
  for (int b = 0; b < B; ++b) {
    const int a = cu[b], e = cu[b + 1];
    for (int q0 = 0; q0 < e - a && n < max_items; q0 += NN_TILE) { NnWork w = {a, e - a, q0, a, e - a, 0, 0, 0}; items[n++] = w; }
  }
  for (; n < max_items; ++n) { NnWork w = {0, 0, 0, 0, 0, 0, 0, 0}; items[n] = w; }
}

// for every query x_i of the work item: d2[i] = min_j |x_i - y_j|^2 over the item's candidates, idx[i] = first arg-min (item-local)
__global__ __launch_bounds__(NN_TILE) void nn_query_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                           const NnWork* __restrict__ items, float* __restrict__ d2_out,
                                                           int32_t* __restrict__ idx_out) {
  __shared__ float4 tile[NN_TILE];
  const NnWork w = items[blockIdx.x];
  if (w.x_len <= 0) return;
  const int q = w.q0 + threadIdx.x;
  const bool active = q < w.x_len;
  const size_t qi = (size_t)w.x_start + (active ? q : w.x_len - 1);
  const float qx = X[qi * 3 + 0], qy = X[qi * 3 + 1], qz = X[qi * 3 + 2];
  float best = __builtin_inff();
  int besti = -1;
  for (int k0 = 0; k0 < w.y_len; k0 += NN_TILE) {
    const int k = k0 + threadIdx.x;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < w.y_len) {
      const size_t ki = (size_t)w.y_start + k;
      c.x = Y[ki * 3 + 0]; c.y = Y[ki * 3 + 1]; c.z = Y[ki * 3 + 2];
    }
    __syncthreads();
    tile[threadIdx.x] = c;
    __syncthreads();
    const int nk = min(NN_TILE, w.y_len - k0);
#pragma unroll 8
    for (int j = 0; j < nk; ++j) {
      const float4 p = tile[j];
      const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
      const float d2 = dx * dx + dy * dy + dz * dz;
      if (d2 < best) { best = d2; besti = k0 + j; }      // strict <: the first minimum wins, as torch.min
    }
  }
  if (active) {
    d2_out[qi] = best;
    if (idx_out) idx_out[qi] = besti;
  }
}

// chamfer RMSE per object: sqrt(0.5 * (mean_i d2_gt->pred[i] + mean_j d2_pred->gt[j]))   (metrics.py:36-43)
__global__ __launch_bounds__(256) void chamfer_finish_kernel(const float* __restrict__ d2a, const float* __restrict__ d2b,
                                                             const int32_t* __restrict__ cu, float* __restrict__ out) {
  __shared__ double red[4];
  const int b = blockIdx.x;
  const int a = cu[b], e = cu[b + 1];
  double s = 0.0;
  for (int i = a + threadIdx.x; i < e; i += 256) s += (double)d2a[i] + (double)d2b[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tot = (red[0] + red[1]) + (red[2] + red[3]);
    out[b] = e > a ? (float)sqrt(0.5 * tot / (double)(e - a)) : __builtin_nanf("");
  }
}

// correspondence RMSE (metrics.py:443-469): valid_i = sqrt(d2_i) <= thr; rmse = sqrt(mean_valid |sp_i - tp_{nn(i)}|^2)
// out[0] = rmse (inf if no correspondence), out[1] = number of correspondences, out[2] = ratio = number / N_source
__global__ __launch_bounds__(256) void correspondence_finish_kernel(const float* __restrict__ d2, const int32_t* __restrict__ nn,
                                                                    const float* __restrict__ sp, const float* __restrict__ tp, int Ns,
                                                                    float thr, float* __restrict__ out) {
  __shared__ double red_s[4];
  __shared__ int red_n[4];
  double s = 0.0;
  int n = 0;
  for (int i = threadIdx.x; i < Ns; i += 256) {
    if (sqrtf(d2[i]) <= thr) {
      const int j = nn[i];
      const float dx = sp[(size_t)i * 3 + 0] - tp[(size_t)j * 3 + 0], dy = sp[(size_t)i * 3 + 1] - tp[(size_t)j * 3 + 1],
                  dz = sp[(size_t)i * 3 + 2] - tp[(size_t)j * 3 + 2];
      s += (double)(dx * dx + dy * dy + dz * dz);
      ++n;
    }
  }
  s = wave_sum_d(s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  if ((threadIdx.x & 63) == 0) { red_s[threadIdx.x >> 6] = s; red_n[threadIdx.x >> 6] = n; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tot = (red_s[0] + red_s[1]) + (red_s[2] + red_s[3]);
    const int cnt = red_n[0] + red_n[1] + red_n[2] + red_n[3];
    out[0] = cnt > 0 ? (float)sqrt(tot / (double)cnt) : __builtin_inff();
    out[1] = (float)cnt;
    out[2] = Ns > 0 ? (float)cnt / (float)Ns : 0.f;
  }
}

size_t nn_max_items(long n, int B) { return (size_t)(n / NN_TILE) + (size_t)B + 1; }

int launch_chamfer_rmse(hipStream_t stream, const float* gt, const float* pred, const int32_t* cu_batch, int B, long TP, float* out,
                        float* d2a, float* d2b, NnWork* items) {
  if (B <= 0 || TP <= 0) return RAP_OK;
  const int max_items = (int)nn_max_items(TP, B);
  hipLaunchKernelGGL(nn_worklist_kernel, dim3(1), dim3(64), 0, stream, cu_batch, B, 0, items, max_items);
  RAP_LAUNCH_CHECK();
  hipLaunchKernelGGL(nn_query_kernel, dim3(max_items), dim3(NN_TILE), 0, stream, gt, pred, items, d2a, (int32_t*)nullptr);
  RAP_LAUNCH_CHECK();
  hipLaunchKernelGGL(nn_query_kernel, dim3(max_items), dim3(NN_TILE), 0, stream, pred, gt, items, d2b, (int32_t*)nullptr);
  RAP_LAUNCH_CHECK();
  hipLaunchKernelGGL(chamfer_finish_kernel, dim3(B), dim3(256), 0, stream, d2a, d2b, cu_batch, out);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

__global__ void nn_pair_worklist_kernel(int Ns, int Nt, NnWork* __restrict__ items, int max_items) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int n = 0;
  for (int q0 = 0; q0 < Ns && n < max_items; q0 += NN_TILE) { NnWork w = {0, Ns, q0, 0, Nt, 0, 0, 0}; items[n++] = w; }
  for (; n < max_items; ++n) { NnWork w = {0, 0, 0, 0, 0, 0, 0, 0}; items[n] = w; }
}

int launch_correspondence_rmse(hipStream_t stream, const float* source_gt, const float* target_gt, const float* source_pred,
                               const float* target_pred, int Ns, int Nt, float thr, float* out3, float* d2, int32_t* nn, NnWork* items) {
  const int max_items = (int)nn_max_items(Ns, 1);
  hipLaunchKernelGGL(nn_pair_worklist_kernel, dim3(1), dim3(64), 0, stream, Ns, Nt, items, max_items);
  RAP_LAUNCH_CHECK();
  hipLaunchKernelGGL(nn_query_kernel, dim3(max_items), dim3(NN_TILE), 0, stream, source_gt, target_gt, items, d2, nn);
  RAP_LAUNCH_CHECK();
  hipLaunchKernelGGL(correspondence_finish_kernel, dim3(1), dim3(256), 0, stream, d2, nn, source_pred, target_pred, Ns, thr, out3);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
