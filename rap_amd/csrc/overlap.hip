// Cross-part overlap ratio on device (the second selection criterion of the reference's test_step,
// modeling.py:594-616): replaces compute_overlap_ratio (rectified_point_flow/eval/metrics.py:625-691), which per sample builds
// chunked (1024 x N) torch.cdist matrices in HBM, masks same-part pairs and takes row minima.
//
//   d_i   = min over points j of the SAME sample in a DIFFERENT part of |x_i - x_j|        (metrics.py:672-679)
//   OR_tau[b] = |{ i in sample b : d_i <= tau }| / N_b ;  0 for samples with <= 1 point or a single non-empty part (:667-668)
//
// The N^2 distance work never touches HBM: a block owns 256 query points of one sample (one per lane, coordinates and part
// id in registers) and streams the sample's points through LDS in tiles of 256 (x, y, z, part id as one float4 -> one
// ds_read_b128 per candidate, broadcast to the whole wave); squared distances by direct differences in fp32 (more accurate
// near the threshold than cdist's |a|^2 + |b|^2 - 2ab expansion), one v_min per pair, sqrt once per query.
// HBM traffic: 16 B per point per query tile of its sample (L2-resident); VALU-bound at ~6 instructions per pair.
#include "kernels.h"

#define OV_TILE 256
struct OverlapWork { int seg_start, seg_len, q0, sample; };

// one thread: work items (sample, 256-query tile); unused slots get seg_len = 0
__global__ void overlap_worklist_kernel(const int32_t* __restrict__ cu, int B, OverlapWork* __restrict__ items, int max_items) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int n = 0;
  for (int b = 0; b < B; ++b) {
    const int a = cu[b], e = cu[b + 1];
    for (int q0 = 0; q0 < e - a && n < max_items; q0 += OV_TILE) { OverlapWork w = {a, e - a, q0, b}; items[n++] = w; }
  }
  for (; n < max_items; ++n) { OverlapWork w = {0, 0, 0, 0}; items[n] = w; }
}

// part id of every point (ppp_to_ids, utils/point_clouds.py:70-92): points are packed (sample, part)-major
__global__ __launch_bounds__(256) void point_part_id_kernel(const int32_t* __restrict__ part_off, int nparts, int P,
                                                            int32_t* __restrict__ pid, long TP) {
  const int part = blockIdx.y;
  const int a = part_off[part], e = part_off[part + 1];
  for (int i = a + blockIdx.x * 256 + threadIdx.x; i < e; i += gridDim.x * 256) pid[i] = part % P;
}

__global__ __launch_bounds__(OV_TILE) void overlap_min_dist_kernel(const float* __restrict__ pts, const int32_t* __restrict__ pid,
                                                                   const OverlapWork* __restrict__ items, float* __restrict__ min_dist) {
  __shared__ float4 tile[OV_TILE];
  const OverlapWork w = items[blockIdx.x];
  if (w.seg_len <= 0) return;
  const int q = w.q0 + threadIdx.x;
  const bool active = q < w.seg_len;
  const int qi = w.seg_start + (active ? q : w.seg_len - 1);
  const float qx = pts[(size_t)qi * 3 + 0], qy = pts[(size_t)qi * 3 + 1], qz = pts[(size_t)qi * 3 + 2];
  const int qp = pid[qi];
  float best = __builtin_inff();
  for (int k0 = 0; k0 < w.seg_len; k0 += OV_TILE) {
    const int k = k0 + threadIdx.x;
    float4 c;
    if (k < w.seg_len) {
      const size_t ki = (size_t)(w.seg_start + k);
      c.x = pts[ki * 3 + 0]; c.y = pts[ki * 3 + 1]; c.z = pts[ki * 3 + 2]; c.w = __int_as_float(pid[ki]);
    } else {
      c.x = c.y = c.z = 0.f; c.w = __int_as_float(-1);                      // padding slots are never read (j < nk)
    }
    __syncthreads();
    tile[threadIdx.x] = c;
    __syncthreads();
    const int nk = min(OV_TILE, w.seg_len - k0);
#pragma unroll 8
    for (int j = 0; j < nk; ++j) {
      const float4 p = tile[j];
      const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
      const float d2 = dx * dx + dy * dy + dz * dz;
      best = (__float_as_int(p.w) != qp) ? fminf(best, d2) : best;           // other parts only (metrics.py:673-676)
    }
  }
  if (active) min_dist[qi] = sqrtf(best);
}

// ratios[t][b] = mean over the sample's points of (min_dist <= tau_t); 0 when N <= 1 or fewer than two non-empty parts
struct OverlapTaus { float tau[8]; int n; };
__global__ __launch_bounds__(256) void overlap_ratio_kernel(const float* __restrict__ min_dist, const int32_t* __restrict__ cu,
                                                            const int32_t* __restrict__ part_off, int P, OverlapTaus taus, int B,
                                                            float* __restrict__ ratios) {
  __shared__ int red[4][8];
  const int b = blockIdx.x;
  const int a = cu[b], e = cu[b + 1];
  int nonempty = 0;
  for (int p = 0; p < P; ++p) nonempty += (part_off[b * P + p + 1] - part_off[b * P + p]) > 0 ? 1 : 0;
  int cnt[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) cnt[t] = 0;
  for (int i = a + threadIdx.x; i < e; i += 256) {
    const float d = min_dist[i];
#pragma unroll
    for (int t = 0; t < 8; ++t) cnt[t] += (t < taus.n && d <= taus.tau[t]) ? 1 : 0;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    int v = cnt[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) red[wave][t] = v;
  }
  __syncthreads();
  if (threadIdx.x < taus.n) {
    const int t = threadIdx.x;
    const int total = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    const int N = e - a;
    ratios[(size_t)t * B + b] = (N <= 1 || nonempty <= 1) ? 0.f : (float)total / (float)N;
  }
}

size_t overlap_max_items(long TP, int B) { return (size_t)(TP / OV_TILE) + (size_t)B + 1; }

int launch_overlap_ratio(hipStream_t stream, const float* pts, const int32_t* cu_batch, const int32_t* part_off, int B, int P,
                         long TP, const float* h_taus, int n_taus, float* ratios, float* min_dist, int32_t* pid, void* items_ws) {
  if (B <= 0 || TP <= 0) return RAP_OK;
  if (n_taus <= 0 || n_taus > 8) return RAP_ERR_INVALID;
  OverlapWork* items = (OverlapWork*)items_ws;
  const int max_items = (int)overlap_max_items(TP, B);
  hipLaunchKernelGGL(overlap_worklist_kernel, dim3(1), dim3(64), 0, stream, cu_batch, B, items, max_items);
  RAP_LAUNCH_CHECK();
  hipLaunchKernelGGL(point_part_id_kernel, dim3(16, B * P), dim3(256), 0, stream, part_off, B * P, P, pid, TP);
  RAP_LAUNCH_CHECK();
  hipLaunchKernelGGL(overlap_min_dist_kernel, dim3(max_items), dim3(OV_TILE), 0, stream, pts, pid, items, min_dist);
  RAP_LAUNCH_CHECK();
  OverlapTaus taus;
  taus.n = n_taus;
  for (int t = 0; t < 8; ++t) taus.tau[t] = t < n_taus ? h_taus[t] : 0.f;
  hipLaunchKernelGGL(overlap_ratio_kernel, dim3(B), dim3(256), 0, stream, min_dist, cu_batch, part_off, P, taus, B, ratios);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
