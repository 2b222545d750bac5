// MiniSpinNet local point-feature extractor on device: the step immediately before
// the sampling path -- it produces the 32-d unit-norm `features` the velocity network consumes.  Replaces
// dataset_process/utils/spinnet/patch_embedder.py:49-183 (MiniSpinNet.forward: select_patches, axis_align [global-z mode],
// normalize, SPT, pnt_layer + max-pool, conv_net, pool_layer), patchnet.py:16-84 (Cylindrical_Net) and
// utils/common.py:230-275, 387-469 (circular pads, voxel grid, sphere_query, var_to_invar), which run pytorch3d ball queries and a
// stack of small cuDNN convolutions with ~60 intermediate tensors per call.
//
// Three stages, nothing but the descriptors leaves the device:
//  1. spin_patch_kernel (one 256-thread block per keypoint): ordered ball query -- the first 512 points (in the caller's shuffled
//     order) with |p - kpt|^2 < r^2, found by a block-wide scan with ballot / popcount prefix so that the index order of
//     pytorch3d.ops.ball_query is preserved; empty slots take the keypoint (patch_embedder.py:122-131); centring on the LAST slot
//     (:142-143) and 1/r normalisation (:185-188) while the patch sits in LDS (6 KB); then the spatial point transformer on the LDS
//     copy: for each of the 420 spherical voxel centres the first 10 patch points within 0.8/3 (utils/common.py:396-440 including its
//     legacy index-0 mask), de-rotated about z by the voxel's azimuth bin (:443-469), pushed through the 3->16 point MLP with
//     folded BatchNorm + ReLU and max-pooled over the 10 slots (patch_embedder.py:75-76) -> (K, 3, 7, 20, 16) channels-last.
//  2. eight convolutions as im2col + the fp32 MFMA GEMM with folded BatchNorm and ReLU in the epilogue: Conv3d 16->64 (circular
//     azimuth pad, zero elevation pad, no radial pad: 3 -> 1) and seven 3x3 Conv2d (64,128,128,64,64,32,32) with the same pads
//     (patchnet.py:49-84, common.py:230-275).  Channels-last activations; output columns are padded to the GEMM's 128-wide tiles.
//  3. spin_pool_kernel (one wave per keypoint): the 1x1 attention pool 32->16->1 with folded BatchNorm + ReLU, weighted mean over
//     the 7x20 map, L2 normalisation (patch_embedder.py:81-83).
#include "kernels.h"
#include "kabsch.h"

#define SP_PATCH 512
#define SP_VOX 420
#define SP_NS 10
#define SP_AZI 20
#define SP_ELE 7

struct SpinConsts {
  float w1[16][3];       // point MLP with BatchNorm folded in
  float b1[16];
};

// ---------------------------------------------------------------------------------------------
// stage 1
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spin_patch_kernel(const float* __restrict__ pts, const int32_t* __restrict__ perm, long N,
                                                         const float* __restrict__ kpts, int K, float des_r,
                                                         const float* __restrict__ vox /* (420,3) */, SpinConsts cw,
                                                         float* __restrict__ x0 /* (K,420,16) */, int lrf) {
  __shared__ float patch[SP_PATCH * 3];
  __shared__ int wave_cnt[4];
  __shared__ int total_s;
  __shared__ double cov_sh[4][6];
  __shared__ float rot_sh[9];
  const int k = blockIdx.x;
  if (k >= K) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float kx = kpts[(size_t)k * 3 + 0], ky = kpts[(size_t)k * 3 + 1], kz = kpts[(size_t)k * 3 + 2];
  const float r2 = des_r * des_r;
  // every slot starts as the keypoint itself (patch_embedder.py:125-131: invalid neighbours are replaced by the reference point)
  for (int i = tid; i < SP_PATCH; i += 256) { patch[i * 3 + 0] = kx; patch[i * 3 + 1] = ky; patch[i * 3 + 2] = kz; }
  if (tid == 0) total_s = 0;
  __syncthreads();
  // ---- ordered ball query: first 512 in-radius points in index order (pytorch3d ball_query: dist2 < radius2)
  for (long base = 0; base < N; base += 256) {
    const int total = total_s;
    if (total >= SP_PATCH) break;
    const long i = base + tid;
    float px = 0.f, py = 0.f, pz = 0.f;
    bool in = false;
    if (i < N) {
      const long src = perm ? (long)perm[i] : i;
      px = pts[src * 3 + 0]; py = pts[src * 3 + 1]; pz = pts[src * 3 + 2];
      const float dx = px - kx, dy = py - ky, dz = pz - kz;
      in = (dx * dx + dy * dy + dz * dz) < r2;
    }
    const unsigned long long m = __ballot(in);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int before = total;
    for (int w = 0; w < wave; ++w) before += wave_cnt[w];
    const int slot = before + __popcll(m & ((1ull << lane) - 1ull));
    if (in && slot < SP_PATCH) { patch[slot * 3 + 0] = px; patch[slot * 3 + 1] = py; patch[slot * 3 + 2] = pz; }
    __syncthreads();
    if (tid == 0) total_s = total + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  // ---- centre on the LAST slot and normalise by the descriptor radius (axis_align global-z mode + normalize)
  const float cx = patch[(SP_PATCH - 1) * 3 + 0], cy = patch[(SP_PATCH - 1) * 3 + 1], cz = patch[(SP_PATCH - 1) * 3 + 2];
  __syncthreads();
  if (lrf) {
    // ---- local reference axis (is_aligned_to_global_z = False; patch_embedder.py:145-151, common.py:472-496, 539-557): z = the
    // singular vector of the smallest singular value of the patch covariance, oriented so that -z . centre >= 0 (the
    // 'normal' disambiguation: towards the sensor at the origin), then the Rodrigues rotation that takes z to (0,0,1).
    double c6[6] = {0, 0, 0, 0, 0, 0};
    for (int i = tid; i < SP_PATCH; i += 256) {
      const double dx = (double)(patch[i * 3 + 0] - cx), dy = (double)(patch[i * 3 + 1] - cy), dz = (double)(patch[i * 3 + 2] - cz);
      c6[0] += dx * dx; c6[1] += dx * dy; c6[2] += dx * dz; c6[3] += dy * dy; c6[4] += dy * dz; c6[5] += dz * dz;
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) { c6[e] = wave_sum_d(c6[e]); if (lane == 0) cov_sh[wave][e] = c6[e]; }
    __syncthreads();
    if (tid == 0) {
      double C[6];
      for (int e = 0; e < 6; ++e) C[e] = cov_sh[0][e] + cov_sh[1][e] + cov_sh[2][e] + cov_sh[3][e];
      double A[3][3] = {{C[0], C[1], C[2]}, {C[1], C[3], C[4]}, {C[2], C[4], C[5]}}, V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
      for (int sweep = 0; sweep < 40; ++sweep) {
        double off = rap_jacobi_pair(A, V, 0, 1);
        off = fmax(off, rap_jacobi_pair(A, V, 0, 2));
        off = fmax(off, rap_jacobi_pair(A, V, 1, 2));
        if (off < 1e-15) break;
      }
      double sg[3];
      for (int j = 0; j < 3; ++j) sg[j] = A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j];
      int jm = 2;                                     // LAPACK orders singular values descending: ties resolve to the last column
      if (sg[1] < sg[jm]) jm = 1;
      if (sg[0] < sg[jm]) jm = 0;
      double zx = V[0][jm], zy = V[1][jm], zz = V[2][jm];        // right singular vector: unit length also when sigma = 0
      if (C[0] + C[3] + C[5] == 0.0) { zx = 0; zy = 0; zz = 1; }  // empty ball (512 copies of the keypoint): svd(0) = I
      if (-(zx * (double)cx + zy * (double)cy + zz * (double)cz) < 0.0) { zx = -zx; zy = -zy; zz = -zz; }
      const double zn = sqrt(zx * zx + zy * zy + zz * zz); zx /= zn; zy /= zn; zz /= zn;
      // c = z x e_z, theta = acos(z . e_z); R = (I + sin K + (1 - cos) K^2) with K = skew(c / |c|); p' = R p
      double ax = zy, ay = -zx, az = 0.0;
      const double an = sqrt(ax * ax + ay * ay);
      if (an > 1e-12) { ax /= an; ay /= an; } else { ax = 0; ay = 0; }      // F.normalize of a zero vector is zero: R = I
      const double ct = fmin(1.0, fmax(-1.0, zz)), st = sqrt(fmax(0.0, 1.0 - ct * ct));
      const double K[3][3] = {{0, -az, ay}, {az, 0, -ax}, {-ay, ax, 0}};
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          double k2 = 0; for (int m = 0; m < 3; ++m) k2 += K[i][m] * K[m][j];
          rot_sh[3 * i + j] = (float)((i == j ? 1.0 : 0.0) + st * K[i][j] + (1.0 - ct) * k2);
        }
    }
    __syncthreads();
  }
  for (int i = tid; i < SP_PATCH; i += 256) {
    float dx = patch[i * 3 + 0] - cx, dy = patch[i * 3 + 1] - cy, dz = patch[i * 3 + 2] - cz;
    if (lrf) {
      const float rx = rot_sh[0] * dx + rot_sh[1] * dy + rot_sh[2] * dz;
      const float ry = rot_sh[3] * dx + rot_sh[4] * dy + rot_sh[5] * dz;
      const float rz = rot_sh[6] * dx + rot_sh[7] * dy + rot_sh[8] * dz;
      dx = rx; dy = ry; dz = rz;
    }
    patch[i * 3 + 0] = dx / des_r;
    patch[i * 3 + 1] = dy / des_r;
    patch[i * 3 + 2] = dz / des_r;
  }
  __syncthreads();
  // ---- spatial point transformer + point MLP + max-pool: voxel centres v = tid, tid + 256
  const float vr = 0.8f / 3.0f;               // delta / rad_n (patch_embedder.py:72)
  const float vr2 = vr * vr;
  for (int v = tid; v < SP_VOX; v += 256) {
    const float vx = vox[v * 3 + 0], vy = vox[v * 3 + 1], vz = vox[v * 3 + 2];
    float sx[SP_NS], sy[SP_NS], sz[SP_NS];
#pragma unroll
    for (int s = 0; s < SP_NS; ++s) { sx[s] = 0.f; sy[s] = 0.f; sz[s] = 0.f; }
    int cnt = 0, first = -1;
    for (int j = 0; j < SP_PATCH && cnt < SP_NS; ++j) {
      const float qx = patch[j * 3 + 0], qy = patch[j * 3 + 1], qz = patch[j * 3 + 2];
      const float dx = vx - qx, dy = vy - qy, dz = vz - qz;
      if (dx * dx + dy * dy + dz * dz < vr2) {
        if (cnt == 0) first = j;
#pragma unroll
        for (int s = 0; s < SP_NS; ++s)
          if (s == cnt) { sx[s] = qx; sy[s] = qy; sz[s] = qz; }
        ++cnt;
      }
    }
    // legacy mask of sphere_query (common.py:418-424): the first sample is dropped when it is patch point 0
    if (first == 0) { sx[0] = 0.f; sy[0] = 0.f; sz[0] = 0.f; }
    // de-rotation about z by the voxel's azimuth bin: p @ Rz(-i * 2 pi / 20)^T  (common.py:443-469)
    const int azi = v % SP_AZI;
    const float ang = -(float)azi * (6.283185307179586f / (float)SP_AZI);
    const float ca = cosf(ang), sa = sinf(ang);
    float best[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) best[c] = -__builtin_inff();
#pragma unroll
    for (int s = 0; s < SP_NS; ++s) {
      const float rx = sx[s] * ca - sy[s] * sa;
      const float ry = sx[s] * sa + sy[s] * ca;
      const float rz = sz[s];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float y = fmaf(cw.w1[c][2], rz, fmaf(cw.w1[c][1], ry, cw.w1[c][0] * rx)) + cw.b1[c];
        best[c] = fmaxf(best[c], fmaxf(y, 0.f));
      }
    }
    float4* dst = reinterpret_cast<float4*>(x0 + ((size_t)k * SP_VOX + v) * 16);
    dst[0] = make_float4(best[0], best[1], best[2], best[3]);
    dst[1] = make_float4(best[4], best[5], best[6], best[7]);
    dst[2] = make_float4(best[8], best[9], best[10], best[11]);
    dst[3] = make_float4(best[12], best[13], best[14], best[15]);
  }
}

// ---------------------------------------------------------------------------------------------
// stage 2: im2col with the cylindrical pads
// ---------------------------------------------------------------------------------------------
// A[(k,h,w)][(ci,dz,dy,dx)] = x0[k][dz][h+dy-1][(w+dx-1) mod 20][ci]   (0 outside the elevation range); row length ldA >= 432
__global__ __launch_bounds__(256) void spin_im2col3d_kernel(const float* __restrict__ x0, int K, float* __restrict__ A, int ldA) {
  const long row = blockIdx.x;                 // (k, h, w)
  const int w = (int)(row % SP_AZI), h = (int)((row / SP_AZI) % SP_ELE);
  const long k = row / (SP_AZI * SP_ELE);
  float* a = A + row * ldA;
  for (int c = threadIdx.x; c < ldA; c += 256) {
    float v = 0.f;
    if (c < 16 * 27) {
      const int ci = c / 27, r = c % 27;
      const int dz = r / 9, dy = (r % 9) / 3, dx = r % 3;
      const int hh = h + dy - 1;
      if (hh >= 0 && hh < SP_ELE) {
        const int ww = (w + dx - 1 + SP_AZI) % SP_AZI;
        v = x0[(((size_t)k * 3 + dz) * (SP_ELE * SP_AZI) + hh * SP_AZI + ww) * 16 + ci];
      }
    }
    a[c] = v;
  }
}
// A[(k,h,w)][(ci,dy,dx)] = y[(k, h+dy-1, (w+dx-1) mod 20)][ci]; y rows have ldy floats (Cin of them meaningful)
__global__ __launch_bounds__(256) void spin_im2col2d_kernel(const float* __restrict__ y, int ldy, int Cin, float* __restrict__ A) {
  const long row = blockIdx.x;
  const int w = (int)(row % SP_AZI), h = (int)((row / SP_AZI) % SP_ELE);
  const long kbase = (row / (SP_AZI * SP_ELE)) * (SP_AZI * SP_ELE);
  float* a = A + row * (size_t)(Cin * 9);
  for (int c = threadIdx.x; c < Cin * 9; c += 256) {
    const int ci = c / 9, r = c % 9;
    const int dy = r / 3, dx = r % 3;
    const int hh = h + dy - 1;
    float v = 0.f;
    if (hh >= 0 && hh < SP_ELE) {
      const int ww = (w + dx - 1 + SP_AZI) % SP_AZI;
      v = y[(size_t)(kbase + hh * SP_AZI + ww) * ldy + ci];
    }
    a[c] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// stage 3: attention pool + L2 normalisation; one wave per keypoint, x rows have ldx floats (32 meaningful)
// ---------------------------------------------------------------------------------------------
struct SpinPoolW { float w1[16][32]; float b1[16]; float w2[16]; float b2; };
__global__ __launch_bounds__(64) void spin_pool_kernel(const float* __restrict__ x, int ldx, int K, const SpinPoolW* __restrict__ pw,
                                                       float* __restrict__ desc) {
  const int k = blockIdx.x, lane = threadIdx.x;
  if (k >= K) return;
  float acc[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) acc[c] = 0.f;
  for (int p = lane; p < SP_ELE * SP_AZI; p += 64) {
    const float* xr = x + ((size_t)k * (SP_ELE * SP_AZI) + p) * ldx;
    float xv[32];
#pragma unroll
    for (int c = 0; c < 32; c += 4) {
      const float4 t = *reinterpret_cast<const float4*>(xr + c);
      xv[c] = t.x; xv[c + 1] = t.y; xv[c + 2] = t.z; xv[c + 3] = t.w;
    }
    float w = pw->b2;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float hsum = pw->b1[j];
#pragma unroll
      for (int c = 0; c < 32; ++c) hsum = fmaf(pw->w1[j][c], xv[c], hsum);
      w = fmaf(pw->w2[j], fmaxf(hsum, 0.f), w);
    }
    w = fmaxf(w, 0.f);
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = fmaf(xv[c], w, acc[c]);
  }
  float nrm2 = 0.f;
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    acc[c] = wave_sum(acc[c]) / (float)(SP_ELE * SP_AZI);       // avg_pool2d over the 7x20 map
    nrm2 = fmaf(acc[c], acc[c], nrm2);
  }
  const float inv = 1.0f / fmaxf(sqrtf(nrm2), 1e-12f);          // F.normalize(p=2, eps=1e-12)
  if (lane < 32) {
    float out = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) out = (lane == c) ? acc[c] * inv : out;
    desc[(size_t)k * 32 + lane] = out;
  }
}

// ---------------------------------------------------------------------------------------------
// weight preparation: fold an eval-mode BatchNorm into the preceding convolution, pad to the GEMM's tile grid
//   W' = W * s, b' = (b - running_mean) * s + beta,  s = gamma / sqrt(running_var + 1e-5)   (gamma = 1, beta = 0 without affine)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spin_fold_kernel(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ rm,
                                                        const float* __restrict__ rv, int Cout, int Kd, float* __restrict__ Wout,
                                                        int ldw, int Npad, float* __restrict__ bout) {
  const int n = blockIdx.x;
  if (n >= Npad) return;
  float s = 1.f, sh = 0.f;
  if (n < Cout && rv) {
    s = (gamma ? gamma[n] : 1.f) / sqrtf(rv[n] + 1e-5f);
    sh = (beta ? beta[n] : 0.f) - rm[n] * s;
  }
  for (int c = threadIdx.x; c < ldw; c += 256) Wout[(size_t)n * ldw + c] = (n < Cout && c < Kd) ? W[(size_t)n * Kd + c] * s : 0.f;
  if (threadIdx.x == 0) bout[n] = n < Cout ? b[n] * s + sh : 0.f;
}

int launch_spin_fold(hipStream_t stream, const float* W, const float* b, const float* gamma, const float* beta, const float* rm,
                     const float* rv, int Cout, int Kd, float* Wout, int ldw, int Npad, float* bout) {
  hipLaunchKernelGGL(spin_fold_kernel, dim3(Npad), dim3(256), 0, stream, W, b, gamma, beta, rm, rv, Cout, Kd, Wout, ldw, Npad, bout);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
int launch_spin_patch(hipStream_t stream, const float* pts, const int32_t* perm, long N, const float* kpts, int K, float des_r,
                      const float* vox, const float* h_w1 /*16x3*/, const float* h_b1 /*16*/, float* x0, int lrf) {
  if (K <= 0) return RAP_OK;
  SpinConsts cw;
  for (int c = 0; c < 16; ++c) { cw.b1[c] = h_b1[c]; for (int d = 0; d < 3; ++d) cw.w1[c][d] = h_w1[c * 3 + d]; }
  hipLaunchKernelGGL(spin_patch_kernel, dim3(K), dim3(256), 0, stream, pts, perm, N, kpts, K, des_r, vox, cw, x0, lrf);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
int launch_spin_im2col3d(hipStream_t stream, const float* x0, int K, float* A, int ldA) {
  if (K <= 0) return RAP_OK;
  hipLaunchKernelGGL(spin_im2col3d_kernel, dim3((unsigned)((long)K * SP_ELE * SP_AZI)), dim3(256), 0, stream, x0, K, A, ldA);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
int launch_spin_im2col2d(hipStream_t stream, const float* y, int ldy, int Cin, int K, float* A) {
  if (K <= 0) return RAP_OK;
  hipLaunchKernelGGL(spin_im2col2d_kernel, dim3((unsigned)((long)K * SP_ELE * SP_AZI)), dim3(256), 0, stream, y, ldy, Cin, A);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
int launch_spin_pool(hipStream_t stream, const float* x, int ldx, int K, const void* d_pool_w, float* desc) {
  if (K <= 0) return RAP_OK;
  hipLaunchKernelGGL(spin_pool_kernel, dim3(K), dim3(64), 0, stream, x, ldx, K, (const SpinPoolW*)d_pool_w, desc);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
