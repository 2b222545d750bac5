// Farthest point sampling on device (surrounding preprocessing): replaces
// pytorch3d.ops.sample_farthest_points as dataset_process/utils/point_sampling_utils.py:263-305 calls it (batched, ragged
// `lengths`, per-cloud K, random start point drawn by the caller) and demo.py:568-571 uses it to pick 200..20 000 keypoints per view.
//
//   selected[0] = start;  d_i = inf;  repeat K-1 times:  d_i = min(d_i, |p_i - p_last|^2);  p_last = FIRST argmax_i d_i
//
// One 1024-thread block per cloud.  The running distances live in the caller's workspace (L2-resident: 4 B per point), the
// coordinates are re-read from L2 every iteration (12 B per point); per iteration each lane folds its strided slice into a
// (distance, index) maximum, the block reduces with wavefront shuffles + one LDS round and broadcasts the winner.
// Inherently sequential in K: ~2-3 us per iteration at 100k points.
#include "kernels.h"

__global__ __launch_bounds__(1024) void fps_kernel(const float* __restrict__ pts, const int32_t* __restrict__ cloud_start /* (C) */,
                                                   const int32_t* __restrict__ cloud_len /* (C) */,
                                                   const int32_t* __restrict__ Ks, const int32_t* __restrict__ starts, int Kmax,
                                                   float* __restrict__ dist /* (total points) */, int32_t* __restrict__ out /* (C,Kmax) */) {
  __shared__ float red_d[16];
  __shared__ int red_i[16];
  __shared__ int winner;
  const int c = blockIdx.x;
  const int a = cloud_start[c], n = cloud_len[c];
  int K = Ks[c];
  K = K < n ? K : n;                                  // pytorch3d: at most `length` points are selected
  int32_t* o = out + (size_t)c * Kmax;
  for (int i = threadIdx.x; i < Kmax; i += 1024) o[i] = -1;      // padding, as pytorch3d
  if (n <= 0 || K <= 0) return;
  const float* p = pts + (size_t)a * 3;
  float* d = dist + a;
  for (int i = threadIdx.x; i < n; i += 1024) d[i] = __builtin_inff();
  int last = starts[c];
  last = last < 0 ? 0 : (last >= n ? n - 1 : last);
  __syncthreads();
  if (threadIdx.x == 0) o[0] = last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 1; k < K; ++k) {
    const float lx = p[(size_t)last * 3 + 0], ly = p[(size_t)last * 3 + 1], lz = p[(size_t)last * 3 + 2];
    float bd = -1.f;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 1024) {
      const float dx = p[(size_t)i * 3 + 0] - lx, dy = p[(size_t)i * 3 + 1] - ly, dz = p[(size_t)i * 3 + 2] - lz;
      const float nd = fminf(d[i], dx * dx + dy * dy + dz * dz);
      d[i] = nd;
      if (nd > bd) { bd = nd; bi = i; }               // strided ascending i: the lane's first maximum
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float od = __shfl_xor(bd, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (od > bd || (od == bd && oi < bi)) { bd = od; bi = oi; }   // first (lowest-index) maximum, as torch.argmax
    }
    if (lane == 0) { red_d[wave] = bd; red_i[wave] = bi; }
    __syncthreads();
    if (wave == 0) {
      float vd = lane < 16 ? red_d[lane] : -2.f;
      int vi = lane < 16 ? red_i[lane] : 0x7fffffff;
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        const float od = __shfl_xor(vd, off, 64);
        const int oi = __shfl_xor(vi, off, 64);
        if (od > vd || (od == vd && oi < vi)) { vd = od; vi = oi; }
      }
      if (lane == 0) { winner = vi; o[k] = vi; }
    }
    __syncthreads();
    last = winner;
  }
}

int launch_fps(hipStream_t stream, const float* pts, const int32_t* cloud_start, const int32_t* cloud_len, const int32_t* Ks,
               const int32_t* starts, int C, int Kmax,
               float* dist, int32_t* out) {
  if (C <= 0 || Kmax <= 0) return RAP_OK;
  hipLaunchKernelGGL(fps_kernel, dim3(C), dim3(1024), 0, stream, pts, cloud_start, cloud_len, Ks, starts, Kmax, dist, out);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
