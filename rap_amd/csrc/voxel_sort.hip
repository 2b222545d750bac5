// Voxel down-sampling / coverage for grids whose dense key table (voxel.hip) would not be worth its size: O(N) memory, a radix sort.
//
// voxel.hip sizes a table by the VOLUME of the grid (v + v^2 + v^3 slots of 8 bytes: a 100 m scene at 5 cm is 64 GB, cleared on every
// call; more than 2^33 slots is refused).  The reference's own formulation (dataset_utils.py:279-322: torch.unique of the keys +
// scatter_reduce_(amin)) scales with the number of POINTS; so does this path, picked by the host wrapper when the table would be
// large against N (rap_amd/point_sampling.py), with identical results:
//   1. one composite 64-bit sort key per point: (voxel key << 10) | quantised centre distance (0..999), payload = point index;
//   2. rocPRIM radix_sort_pairs -- stable, so equal (voxel, level) pairs stay in index order;
//   3. the first element of every voxel run is the reference's winner (smallest level, ties to the lowest index); run heads are
//      flagged, scanned and emitted in ascending key order -- the order the dense path and torch.unique produce.
// rocPRIM is the device-primitives library of the ROCm platform itself (no CUDA layer); sorting is plumbing around the hot path.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "kernels.h"

// the arithmetic of voxel_of / voxel_fill_kernel in voxel.hip, restated (same rounding: no FMA contraction)
__global__ __launch_bounds__(256) void voxel_sort_keys_kernel(const float* __restrict__ pts, long N, float vs, long long ox, long long oy,
                                                              long long oz, long long v, float dmax, unsigned long long* __restrict__ keys,
                                                              unsigned int* __restrict__ vals) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long)gridDim.x * 256) {
    const float* p = pts + i * 3;
    const float fx = floorf(p[0] / vs), fy = floorf(p[1] / vs), fz = floorf(p[2] / vs);
    const float cx = mul_rn_nofuse(fx + 0.5f, vs), cy = mul_rn_nofuse(fy + 0.5f, vs), cz = mul_rn_nofuse(fz + 0.5f, vs);
    const float dx = p[0] - cx, dy = p[1] - cy, dz = p[2] - cz;
    float s = mul_rn_nofuse(dx, dx);
    s = s + mul_rn_nofuse(dy, dy);
    s = s + mul_rn_nofuse(dz, dz);
    const float d = sqrtf(s);
    const unsigned long long lvl = (unsigned long long)(long long)(mul_rn_nofuse(d / dmax, 999.0f));
    const long long key = ((long long)fx - ox) + ((long long)fy - oy) * v + ((long long)fz - oz) * v * v;
    keys[i] = ((unsigned long long)key << 10) | (lvl & 1023ull);
    vals[i] = (unsigned int)i;
  }
}
// exact (collision-free) voxel id for the coverage count: per-axis extents as strides
__global__ __launch_bounds__(256) void voxel_sort_exact_keys_kernel(const float* __restrict__ pts, long N, float vs, long long ox, long long oy,
                                                                    long long oz, long long ex, long long ey,
                                                                    unsigned long long* __restrict__ keys) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long)gridDim.x * 256) {
    const float* p = pts + i * 3;
    const long long gx = (long long)floorf(p[0] / vs) - ox, gy = (long long)floorf(p[1] / vs) - oy, gz = (long long)floorf(p[2] / vs) - oz;
    keys[i] = (unsigned long long)(gx + ex * (gy + ey * gz));
  }
}
__global__ __launch_bounds__(256) void voxel_sort_heads_kernel(const unsigned long long* __restrict__ keys, long N, int shift,
                                                               unsigned int* __restrict__ flags) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < N) flags[i] = (i == 0 || (keys[i] >> shift) != (keys[i - 1] >> shift)) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void voxel_sort_emit_kernel(const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ vals,
                                                              const unsigned int* __restrict__ pos, long N, long long* __restrict__ idx_out,
                                                              int* __restrict__ count_out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const bool head = i == 0 || (keys[i] >> 10) != (keys[i - 1] >> 10);
  if (head) idx_out[pos[i] - 1] = (long long)vals[i];      // pos = inclusive scan of the head flags
  if (i == N - 1) count_out[0] = (int)pos[i];
}
__global__ void voxel_sort_count_kernel(const unsigned int* __restrict__ pos, long N, long long* __restrict__ count_out) {
  count_out[0] = (long long)pos[N - 1];
}

struct VoxelSortWs {
  unsigned long long *keys_a, *keys_b;
  unsigned int *vals_a, *vals_b, *flags, *pos;
  void* temp;
  size_t temp_bytes, total;
};
static size_t vs_align(size_t x) { return (x + 255) & ~(size_t)255; }
static int voxel_sort_carve(long N, char* base, VoxelSortWs& w) {
  size_t t_sort = 0, t_keys = 0, t_scan = 0;
  if (rocprim::radix_sort_pairs(nullptr, t_sort, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned int*)nullptr,
                                (unsigned int*)nullptr, (size_t)N, 0, 64, (hipStream_t)0) != hipSuccess)
    return RAP_ERR_HIP;
  if (rocprim::radix_sort_keys(nullptr, t_keys, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (size_t)N, 0, 64, (hipStream_t)0) !=
      hipSuccess)
    return RAP_ERR_HIP;
  if (rocprim::inclusive_scan(nullptr, t_scan, (unsigned int*)nullptr, (unsigned int*)nullptr, (size_t)N, rocprim::plus<unsigned int>(),
                              (hipStream_t)0) != hipSuccess)
    return RAP_ERR_HIP;
  w.temp_bytes = t_sort > t_keys ? t_sort : t_keys;
  w.temp_bytes = w.temp_bytes > t_scan ? w.temp_bytes : t_scan;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += vs_align(bytes); return p; };
  w.keys_a = (unsigned long long*)take((size_t)N * 8);
  w.keys_b = (unsigned long long*)take((size_t)N * 8);
  w.vals_a = (unsigned int*)take((size_t)N * 4);
  w.vals_b = (unsigned int*)take((size_t)N * 4);
  w.flags = (unsigned int*)take((size_t)N * 4);
  w.pos = (unsigned int*)take((size_t)N * 4);
  w.temp = take(w.temp_bytes + 256);
  w.total = off;
  return RAP_OK;
}
size_t voxel_sorted_workspace_bytes(long N) {
  VoxelSortWs w;
  return (N > 0 && voxel_sort_carve(N, nullptr, w) == RAP_OK) ? w.total : 0;
}

static unsigned vs_grid(long N) { return (unsigned)((N + 255) / 256 < 4096 ? (N + 255) / 256 : 4096); }

int launch_voxel_downsample_sorted(hipStream_t stream, const float* pts, long N, float vs, const long long* h_bounds6, float dmax, void* ws,
                                   size_t ws_bytes, long long* idx_out, int* count_out) {
  long long v = 0;
  for (int a = 0; a < 3; ++a) v = (h_bounds6[3 + a] - h_bounds6[a]) > v ? (h_bounds6[3 + a] - h_bounds6[a]) : v;
  if (v >= (1LL << 18)) return RAP_ERR_INVALID;            // key < 2^54 so that (key << 10 | level) fits 64 bits: v + v^2 + v^3 < 2^54
  VoxelSortWs w;
  int rc = voxel_sort_carve(N, (char*)ws, w);
  if (rc) return rc;
  if (w.total > ws_bytes) return RAP_ERR_WORKSPACE;
  hipLaunchKernelGGL(voxel_sort_keys_kernel, dim3(vs_grid(N)), dim3(256), 0, stream, pts, N, vs, h_bounds6[0], h_bounds6[1], h_bounds6[2], v, dmax,
                     w.keys_a, w.vals_a);
  RAP_LAUNCH_CHECK();
  size_t tb = w.temp_bytes;
  RAP_HIP_CHECK(rocprim::radix_sort_pairs(w.temp, tb, w.keys_a, w.keys_b, w.vals_a, w.vals_b, (size_t)N, 0, 64, stream));
  const unsigned nb = (unsigned)((N + 255) / 256);
  hipLaunchKernelGGL(voxel_sort_heads_kernel, dim3(nb), dim3(256), 0, stream, w.keys_b, N, 10, w.flags);
  RAP_LAUNCH_CHECK();
  tb = w.temp_bytes;
  RAP_HIP_CHECK(rocprim::inclusive_scan(w.temp, tb, w.flags, w.pos, (size_t)N, rocprim::plus<unsigned int>(), stream));
  hipLaunchKernelGGL(voxel_sort_emit_kernel, dim3(nb), dim3(256), 0, stream, w.keys_b, w.vals_b, w.pos, N, idx_out, count_out);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

int launch_voxel_coverage_sorted(hipStream_t stream, const float* pts, long N, float vs, const long long* h_bounds6, void* ws, size_t ws_bytes,
                                 long long* count_out) {
  const __int128 ex = h_bounds6[3] - h_bounds6[0] + 1, ey = h_bounds6[4] - h_bounds6[1] + 1, ez = h_bounds6[5] - h_bounds6[2] + 1;
  if (ex <= 0 || ey <= 0 || ez <= 0 || ex * ey * ez >= ((__int128)1 << 63)) return RAP_ERR_INVALID;
  VoxelSortWs w;
  int rc = voxel_sort_carve(N, (char*)ws, w);
  if (rc) return rc;
  if (w.total > ws_bytes) return RAP_ERR_WORKSPACE;
  hipLaunchKernelGGL(voxel_sort_exact_keys_kernel, dim3(vs_grid(N)), dim3(256), 0, stream, pts, N, vs, h_bounds6[0], h_bounds6[1], h_bounds6[2],
                     (long long)ex, (long long)ey, w.keys_a);
  RAP_LAUNCH_CHECK();
  size_t tb = w.temp_bytes;
  RAP_HIP_CHECK(rocprim::radix_sort_keys(w.temp, tb, w.keys_a, w.keys_b, (size_t)N, 0, 64, stream));
  const unsigned nb = (unsigned)((N + 255) / 256);
  hipLaunchKernelGGL(voxel_sort_heads_kernel, dim3(nb), dim3(256), 0, stream, w.keys_b, N, 0, w.flags);
  RAP_LAUNCH_CHECK();
  tb = w.temp_bytes;
  RAP_HIP_CHECK(rocprim::inclusive_scan(w.temp, tb, w.flags, w.pos, (size_t)N, rocprim::plus<unsigned int>(), stream));
  hipLaunchKernelGGL(voxel_sort_count_kernel, dim3(1), dim3(1), 0, stream, w.pos, N, count_out);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
