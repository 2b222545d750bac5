// Input side of the sampling boundary: raw multi-part scans -> the packed batch the sampler consumes.
//
// Replaces, in their evaluation-split form (no rotation / scale augmentation), the reference's numpy
//   PointCloudDataset._transform        rectified_point_flow/data/dataset.py:733-900
//   variable_collate_fn                 rectified_point_flow/data/datamodule.py:169-198
// for a whole batch in three launches, with nothing read back to the host:
//   per sample   tran_global = mean of all points (:759); primary part = first arg-max of the point counts (:764); primary_trans =
//                its centroid (:769); scale = 1.5 * max |p - primary_trans| over the primary part (:783);
//                pts_gt = (p - primary_trans) / scale - gt_trans, gt_trans = mean of the scaled cloud (:791-796)
//   per part     trans_i = centroid of the part in that frame, cond = pts_gt - trans_i, shuffled inside the part with the caller's
//                permutation (:802-830; np.random.permutation in the reference); the anchor (= primary) part keeps its pose:
//                cond = pts_gt + gt_trans, trans = -gt_trans (:860-870); rotations = I, zero rows for padded parts (pad_data)
// Arithmetic is fp64 like numpy's and cast to fp32 at the end, so the result differs from the reference's only through the
// summation order of the means (1e-16 relative) -- far below one fp32 rounding.
// Kernels: HBM-bound, 12 or 24 B read per point per pass, three passes over the points (part sums, primary-part extent, apply).
#include "kernels.h"

#define COL_THREADS 256

__device__ __forceinline__ double3 load_pt(const void* pts, int f64, long i) {
  if (f64) { const double* p = reinterpret_cast<const double*>(pts) + 3 * i; return make_double3(p[0], p[1], p[2]); }
  const float* p = reinterpret_cast<const float*>(pts) + 3 * i;
  return make_double3((double)p[0], (double)p[1], (double)p[2]);
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < COL_THREADS / 64; ++w) r += sh[w];
  return r;
}

// one block per part: fp64 coordinate sums
__global__ __launch_bounds__(COL_THREADS) void collate_part_sums_kernel(const void* __restrict__ pts, int f64,
                                                                       const int32_t* __restrict__ off, double* __restrict__ psum) {
  __shared__ double sh[COL_THREADS / 64];
  const int part = blockIdx.x;
  const int a = off[part], b = off[part + 1];
  double sx = 0, sy = 0, sz = 0;
  for (int i = a + threadIdx.x; i < b; i += COL_THREADS) {
    const double3 p = load_pt(pts, f64, i);
    sx += p.x; sy += p.y; sz += p.z;
  }
  sx = block_sum(sx, sh); sy = block_sum(sy, sh); sz = block_sum(sz, sh);
  if (threadIdx.x == 0) { psum[3 * part + 0] = sx; psum[3 * part + 1] = sy; psum[3 * part + 2] = sz; }
}

// one block per sample: frame of the sample and the per-part poses.
// frame[b] = {primary_trans xyz, scale, gt_trans xyz, primary part index}; ptrans[part] = centroid of the part in the final frame (fp64)
__global__ __launch_bounds__(COL_THREADS) void collate_frame_kernel(const void* __restrict__ pts, int f64, const int32_t* __restrict__ off,
                                                                   const double* __restrict__ psum, int P, double* __restrict__ frame,
                                                                   double* __restrict__ ptrans, float* __restrict__ rotations,
                                                                   float* __restrict__ translations, float* __restrict__ scales,
                                                                   uint8_t* __restrict__ anchor_parts, float* __restrict__ global_translation,
                                                                   int64_t* __restrict__ cu_seqlens) {
  __shared__ double sh[COL_THREADS / 64];
  __shared__ int s_primary;
  const int b = blockIdx.x;
  const int p0 = b * P;
  if (threadIdx.x == 0) {
    int best = 0, bestn = -1;
    for (int i = 0; i < P; ++i) { const int n = off[p0 + i + 1] - off[p0 + i]; if (n > bestn) { bestn = n; best = i; } }   // first arg-max
    s_primary = best;
    cu_seqlens[b + 1] = (int64_t)off[p0 + P];
    if (b == 0) cu_seqlens[0] = 0;
  }
  __syncthreads();
  const int primary = s_primary;
  const int a = off[p0 + primary], e = off[p0 + primary + 1];
  const int n_all = off[p0 + P] - off[p0];
  double tx = 0, ty = 0, tz = 0;
  for (int i = 0; i < P; ++i) { tx += psum[3 * (p0 + i)]; ty += psum[3 * (p0 + i) + 1]; tz += psum[3 * (p0 + i) + 2]; }
  const double inv_all = n_all > 0 ? 1.0 / (double)n_all : 0.0;
  const double gx = tx * inv_all, gy = ty * inv_all, gz = tz * inv_all;                 // tran_global (original units)
  const double inv_p = e > a ? 1.0 / (double)(e - a) : 0.0;
  const double px = psum[3 * (p0 + primary)] * inv_p, py = psum[3 * (p0 + primary) + 1] * inv_p, pz = psum[3 * (p0 + primary) + 2] * inv_p;
  double m = 0.0;
  for (int i = a + threadIdx.x; i < e; i += COL_THREADS) {
    const double3 p = load_pt(pts, f64, i);
    m = fmax(m, fmax(fabs(p.x - px), fmax(fabs(p.y - py), fabs(p.z - pz))));
  }
  // block max through the sum helper's staging array (max is exact, order-free)
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
  const double scale = m * 1.5;
  const double inv_s = scale > 0.0 ? 1.0 / scale : 0.0;
  // gt_trans = mean((p - primary_trans) / scale)
  const double qx = (gx - px) / scale, qy = (gy - py) / scale, qz = (gz - pz) / scale;
  (void)inv_s;
  if (threadIdx.x == 0) {
    double* f = frame + 8 * b;
    f[0] = px; f[1] = py; f[2] = pz; f[3] = scale; f[4] = qx; f[5] = qy; f[6] = qz; f[7] = (double)primary;
    scales[b] = (float)scale;
    global_translation[3 * b + 0] = (float)gx; global_translation[3 * b + 1] = (float)gy; global_translation[3 * b + 2] = (float)gz;
  }
  if ((int)threadIdx.x < P) {                           // P <= 256 = COL_THREADS (checked by rap_collate_transform)
    const int i = threadIdx.x, part = p0 + i;
    const int n = off[part + 1] - off[part];
    double cx = 0, cy = 0, cz = 0;
    if (n > 0) {
      const double inv = 1.0 / (double)n;
      cx = (psum[3 * part] * inv - px) / scale - qx; cy = (psum[3 * part + 1] * inv - py) / scale - qy; cz = (psum[3 * part + 2] * inv - pz) / scale - qz;
    }
    ptrans[3 * part] = cx; ptrans[3 * part + 1] = cy; ptrans[3 * part + 2] = cz;
    const bool anch = i == primary;
    float* R = rotations + 9 * (size_t)part; float* t = translations + 3 * (size_t)part;
    for (int k = 0; k < 9; ++k) R[k] = (n > 0 && (k == 0 || k == 4 || k == 8)) ? 1.f : 0.f;
    t[0] = n > 0 ? (anch ? (float)(-qx) : (float)cx) : 0.f;
    t[1] = n > 0 ? (anch ? (float)(-qy) : (float)cy) : 0.f;
    t[2] = n > 0 ? (anch ? (float)(-qz) : (float)cz) : 0.f;
    anchor_parts[part] = anch ? 1 : 0;
  }
}

// one thread per output point
__global__ __launch_bounds__(COL_THREADS) void collate_apply_kernel(const void* __restrict__ pts, int f64, const int32_t* __restrict__ off,
                                                                   int nparts, int P, const int64_t* __restrict__ order,
                                                                   const double* __restrict__ frame, const double* __restrict__ ptrans,
                                                                   const float* __restrict__ feat_in, int F, float* __restrict__ cond,
                                                                   float* __restrict__ gt, float* __restrict__ feat_out,
                                                                   uint8_t* __restrict__ anchor_idx, int64_t* __restrict__ part_idx, long TP) {
  const long j = (long)blockIdx.x * COL_THREADS + threadIdx.x;
  if (j >= TP) return;
  // part of point j: last part whose start is <= j (empty parts share their start with the next one: skip them)
  int lo = 0, hi = nparts;              // invariant: off[lo] <= j < off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= j) lo = mid; else hi = mid;
  }
  const int part = lo, b = part / P, i = part - b * P;
  const int a = off[part];
  const long src = order ? (long)a + (long)order[j] : j;
  const double* f = frame + 8 * b;
  const double3 p = load_pt(pts, f64, src);
  const double scale = f[3];
  const double gxs = (p.x - f[0]) / scale - f[4], gys = (p.y - f[1]) / scale - f[5], gzs = (p.z - f[2]) / scale - f[6];
  const bool anch = i == (int)f[7];
  gt[3 * j + 0] = (float)gxs; gt[3 * j + 1] = (float)gys; gt[3 * j + 2] = (float)gzs;
  if (anch) {
    cond[3 * j + 0] = (float)(gxs + f[4]); cond[3 * j + 1] = (float)(gys + f[5]); cond[3 * j + 2] = (float)(gzs + f[6]);
  } else {
    const double* c = ptrans + 3 * part;
    cond[3 * j + 0] = (float)(gxs - c[0]); cond[3 * j + 1] = (float)(gys - c[1]); cond[3 * j + 2] = (float)(gzs - c[2]);
  }
  anchor_idx[j] = anch ? 1 : 0;
  part_idx[j] = i;
  if (F > 0) {
    const float* fi = feat_in + (size_t)src * F;
    float* fo = feat_out + (size_t)j * F;
    for (int k = 0; k < F; ++k) fo[k] = fi[k];
  }
}

// within-part permutation must be a permutation of [0, n_part): checked on the device (a wrong index would read another part)
__global__ __launch_bounds__(COL_THREADS) void collate_check_order_kernel(const int32_t* __restrict__ off, int nparts, const int64_t* __restrict__ order,
                                                                         long TP, int32_t* __restrict__ flag) {
  const long j = (long)blockIdx.x * COL_THREADS + threadIdx.x;
  if (j >= TP) return;
  int lo = 0, hi = nparts;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= j) lo = mid; else hi = mid; }
  const long n = off[lo + 1] - off[lo];
  if (order[j] < 0 || order[j] >= n) atomicOr(flag, 1);
}

size_t collate_workspace_bytes(int B, int P) {
  const size_t np = (size_t)B * P;
  return align_up((np + 1) * 4, 256) + align_up(np * 3 * 8, 256) + align_up((size_t)B * 8 * 8, 256) + align_up(np * 3 * 8, 256) + 256;
}

int launch_collate_transform(hipStream_t stream, const void* pts, int f64, const int64_t* points_per_part, int B, int P, long TP,
                             const int64_t* order, const float* feat_in, int F, float* cond, float* gt, float* feat_out,
                             uint8_t* anchor_idx, int64_t* part_idx, float* rotations, float* translations, float* scales,
                             uint8_t* anchor_parts, float* global_translation, int64_t* cu_seqlens, int32_t* order_flag, void* ws) {
  const int np = B * P;
  char* w = (char*)ws;
  int32_t* off = (int32_t*)w; w += align_up(((size_t)np + 1) * 4, 256);
  double* psum = (double*)w; w += align_up((size_t)np * 3 * 8, 256);
  double* frame = (double*)w; w += align_up((size_t)B * 8 * 8, 256);
  double* ptrans = (double*)w;
  int rc;
  if ((rc = launch_part_offsets(stream, points_per_part, np, off))) return rc;
  hipLaunchKernelGGL(collate_part_sums_kernel, dim3(np), dim3(COL_THREADS), 0, stream, pts, f64, off, psum);
  RAP_LAUNCH_CHECK();
  hipLaunchKernelGGL(collate_frame_kernel, dim3(B), dim3(COL_THREADS), 0, stream, pts, f64, off, psum, P, frame, ptrans, rotations,
                     translations, scales, anchor_parts, global_translation, cu_seqlens);
  RAP_LAUNCH_CHECK();
  if (TP > 0) {
    const unsigned grid = (unsigned)((TP + COL_THREADS - 1) / COL_THREADS);
    if (order && order_flag) {
      hipLaunchKernelGGL(collate_check_order_kernel, dim3(grid), dim3(COL_THREADS), 0, stream, off, np, order, TP, order_flag);
      RAP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(collate_apply_kernel, dim3(grid), dim3(COL_THREADS), 0, stream, pts, f64, off, np, P, order, frame, ptrans, feat_in, F,
                       cond, gt, feat_out, anchor_idx, part_idx, TP);
    RAP_LAUNCH_CHECK();
  }
  return RAP_OK;
}
