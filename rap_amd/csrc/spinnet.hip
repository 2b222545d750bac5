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
  // ---- ordered ball query: first 512 in-radius points in index order (pytorch3d ball_query: dist2 < radius2).
  // 1024 candidates per block iteration, 4 CONSECUTIVE indices per thread (r02: a quarter of the block-wide barriers of the
  // one-point-per-thread scan); a thread's hits go to slots before + (hits of lower lanes) + (own earlier hits), where the lane prefix of
  // the per-thread counts 0..4 comes from three ballots (one per bit of the count).
  for (long base = 0; base < N; base += 1024) {
    const int total = total_s;
    if (total >= SP_PATCH) break;
    float px[4], py[4], pz[4];
    bool in[4];
    int cnt = 0;
    if (!perm && base + tid * 4 + 3 < N) {
      // the caller's order IS the scan order (the host mirror gathers pts[perm] once): 4 consecutive points = 48 contiguous bytes
      const float4* q4 = reinterpret_cast<const float4*>(pts + (base + tid * 4) * 3);
      const float4 u0 = q4[0], u1 = q4[1], u2 = q4[2];
      px[0] = u0.x; py[0] = u0.y; pz[0] = u0.z; px[1] = u0.w; py[1] = u1.x; pz[1] = u1.y;
      px[2] = u1.z; py[2] = u1.w; pz[2] = u2.x; px[3] = u2.y; py[3] = u2.z; pz[3] = u2.w;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float dx = px[j] - kx, dy = py[j] - ky, dz = pz[j] - kz;
        in[j] = (dx * dx + dy * dy + dz * dz) < r2;
        cnt += in[j] ? 1 : 0;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long i = base + tid * 4 + j;
        in[j] = false; px[j] = py[j] = pz[j] = 0.f;
        if (i < N) {
          const long src = perm ? (long)perm[i] : i;
          px[j] = pts[src * 3 + 0]; py[j] = pts[src * 3 + 1]; pz[j] = pts[src * 3 + 2];
          const float dx = px[j] - kx, dy = py[j] - ky, dz = pz[j] - kz;
          in[j] = (dx * dx + dy * dy + dz * dz) < r2;
        }
        cnt += in[j] ? 1 : 0;
      }
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned long long m0 = __ballot(cnt & 1), m1 = __ballot(cnt & 2), m2 = __ballot(cnt & 4);
    const int lane_prefix = __popcll(m0 & lt) + 2 * __popcll(m1 & lt) + 4 * __popcll(m2 & lt);
    if (lane == 0) wave_cnt[wave] = __popcll(m0) + 2 * __popcll(m1) + 4 * __popcll(m2);
    __syncthreads();
    int slot = total + lane_prefix;
    for (int w = 0; w < wave; ++w) slot += wave_cnt[w];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (in[j]) {
        if (slot < SP_PATCH) { patch[slot * 3 + 0] = px[j]; patch[slot * 3 + 1] = py[j]; patch[slot * 3 + 2] = pz[j]; }
        ++slot;
      }
    }
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
// stage 2 (r02): the seven 3x3 cylindrical convolutions as IMPLICIT GEMMs on the fp32 matrix cores -- no im2col round trip (r01: the
// materialised (K*140, 9 Cin) matrix cost 3.4 of the 8 ms per 2048-keypoint chunk) and no padding of the 32- / 64-channel layers to 128
// output columns (r01: 4x / 2x the MFMA work on those layers).
//   C[(k,h,w)][n] = bias[n] + sum_{tap=(dy,dx)} sum_ci  y[k][h+dy-1][(w+dx-1) mod 20][ci] * Wt[n][tap*Cin + ci]      (+ ReLU)
// Structure = gemm_f32_dma_kernel (LDS-DMA tiles, XOR slot swizzle, v_mfma_f32_32x32x2_f32, two LDS stages, two blocks per CU): the
// contraction index is ordered tap-major, so a 32-float k-tile lies inside ONE tap (Cin is a multiple of 32) and the A row piece of a
// pixel is 128 contiguous bytes of the shifted input pixel -- the per-lane DMA source address does the im2col gather; rows outside the
// elevation range (zero pad) read a zero page.  Activations are channels-last with exactly Cin / Cout floats per pixel.
// Block tile (64 WM) x (32 TN WN): 128 x 128 for the 128-channel layers, 256 x 64 and 256 x 32 for the 64- / 32-channel ones.
// ---------------------------------------------------------------------------------------------
struct SpinConvParams {
  const float* y; int ldy; int Cin;   // input (M, ldy) channels-last (Cin meaningful floats per pixel), M = K * 140 pixels
  const float* Wt;                // (Cout, 9 * Cin) tap-major, BatchNorm folded
  const float* bias;              // (Cout)
  const float* zeros;             // >= 128 bytes of zeros
  float* out; int Cout;           // output (M, Cout)
  int M;
};
// C3D: the first layer, Conv3d 16 -> 64 over the (3 radial, 7 elevation, 20 azimuth) grid with no radial pad (3 -> 1): 27 taps x 16 input channels =
// 432 contraction values, padded to 14 k-tiles; a 16-byte DMA chunk (4 channels) lies inside one tap, so every lane gathers its own chunk.
template <int WM, int WN, int TN, int RELU, int C3D = 0>
__global__ __launch_bounds__(256, 2) void spin_conv3x3_kernel(SpinConvParams p) {
  constexpr int BM = 64 * WM, BN = 32 * TN * WN;
  constexpr int CA = BM * 8 / 256, CB = (BN * 8 + 255) / 256;
  constexpr int AF = BM * 32, BF = BN * 32;                     // floats per A / B stage
  static_assert(WM * WN == 4, "four waves");
  extern __shared__ __attribute__((aligned(1024))) float smem_c[];   // [A0 A1 B0 B1]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM;                                // BN == Cout: one n-tile

  // A chunk i of this thread: LDS row (i*256 + tid) >> 3 = pixel m0 + row, slot (id & 7) holding logical slot (slot ^ swz(row))
  int a_pix[CA], a_h[CA], a_w[CA], a_ls[CA];
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int id = i * 256 + tid;
    const int row = id >> 3;
    int m = m0 + row; m = m < p.M ? m : p.M - 1;
    a_ls[i] = 4 * ((id & 7) ^ ((row >> 1) & 7));
    a_w[i] = m % SP_AZI; a_h[i] = (m / SP_AZI) % SP_ELE; a_pix[i] = m - a_h[i] * SP_AZI - a_w[i];      // first pixel of the keypoint
    if (C3D) a_pix[i] *= 3;                                                                            // 420 input voxels per keypoint
  }
  const float* w_src[CB];
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    const int id = i * 256 + tid;
    const int row = (id >> 3) < BN ? (id >> 3) : BN - 1;
    w_src[i] = p.Wt + (size_t)row * (C3D ? 448 : 9 * p.Cin) + 4 * ((id & 7) ^ ((row >> 1) & 7));
  }

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = C3D ? 14 : 9 * p.Cin / 32;
  const int tiles_per_tap = C3D ? 1 : p.Cin / 32;
  const int sw = (l31 >> 1) & 7;
  const int a_row = (wm * 64 + l31) * 32;
  const int b_row = 2 * AF + (wn * 32 * TN + l31) * 32;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem_c;
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
#define SC_DMA1(GSRC, LDSB)                                                                                   \
  {                                                                                                           \
    unsigned keep_;                                                                                           \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(GSRC), "s"(LDSB) : "memory");                                           \
  }
  auto dma = [&](int kt, int buf) {
    const int tap = kt / tiles_per_tap, c0 = (kt - tap * tiles_per_tap) * 32;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const float* src;
      if (C3D) {
        const int kidx = kt * 32 + a_ls[i];                       // this lane's 4 contraction values: one tap, 4 channels
        const int t3 = kidx >> 4, ci0 = kidx & 15;
        const int dz = t3 / 9, r9 = t3 - 9 * dz;
        const int hh = a_h[i] + r9 / 3 - 1;
        int ww = a_w[i] + r9 % 3 - 1; ww = ww < 0 ? ww + SP_AZI : (ww >= SP_AZI ? ww - SP_AZI : ww);
        src = (t3 < 27 && hh >= 0 && hh < SP_ELE) ? p.y + (size_t)(a_pix[i] + dz * (SP_ELE * SP_AZI) + hh * SP_AZI + ww) * 16 + ci0 : p.zeros;
      } else {
        const int hh = a_h[i] + dy;
        int ww = a_w[i] + dx; ww = ww < 0 ? ww + SP_AZI : (ww >= SP_AZI ? ww - SP_AZI : ww);
        src = (hh >= 0 && hh < SP_ELE) ? p.y + (size_t)(a_pix[i] + hh * SP_AZI + ww) * p.ldy + c0 + a_ls[i] : p.zeros + a_ls[i];
      }
      SC_DMA1(src, lds_wave + (unsigned)((buf * AF + i * 1024) * 4))
    }
#pragma unroll
    for (int i = 0; i < CB; ++i)
      if (CB * 256 <= BN * 8 || i * 256 + tid < BN * 8)            // BN = 32: only the first 256 chunks exist
        SC_DMA1(w_src[i] + (size_t)kt * 32, lds_wave + (unsigned)((2 * AF + buf * BF + i * 1024) * 4))
  };
#define SC_SYNC asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
#define SC_RD(OFF) (*reinterpret_cast<const float4*>(&smem_c[OFF]))
  struct Frag { float4 a0, a1, b[TN]; } f0, f1;
  auto read_frag = [&](Frag& f, int buf, int g) {
    const int co = ((2 * g + hi) ^ sw) * 4;
    f.a0 = SC_RD(buf * AF + a_row + co);
    f.a1 = SC_RD(buf * AF + a_row + 32 * 32 + co);
#pragma unroll
    for (int j = 0; j < TN; ++j) f.b[j] = SC_RD(buf * BF + b_row + j * 32 * 32 + co);
  };
#define SC_STEP(F, C)                                                                                        \
  _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                           \
    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a0.C, F.b[j].C, acc[0][j], 0, 0, 0);                   \
    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a1.C, F.b[j].C, acc[1][j], 0, 0, 0);                   \
  }
#define SC_MFMA(F) SC_STEP(F, x) SC_STEP(F, y) SC_STEP(F, z) SC_STEP(F, w)
#define SC_FENCE __builtin_amdgcn_sched_barrier(0);

  dma(0, 0);
  SC_SYNC
  read_frag(f0, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) dma(kt + 1, cur ^ 1);
    read_frag(f1, cur, 1);
    SC_FENCE
    SC_MFMA(f0)
    SC_FENCE
    read_frag(f0, cur, 2);
    SC_FENCE
    SC_MFMA(f1)
    SC_FENCE
    read_frag(f1, cur, 3);
    SC_FENCE
    SC_MFMA(f0)
    SC_FENCE
    SC_SYNC                                   // tile kt+1 landed and visible; every wave is done reading tile kt
    if (more) read_frag(f0, cur ^ 1, 0);
    SC_FENCE
    SC_MFMA(f1)
    SC_FENCE
  }
  // epilogue: lane = output channel, registers = pixels (32 consecutive channels per store instruction: 128-byte row pieces)
  const int mw = m0 + wm * 64, nw = wn * 32 * TN;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = nw + 32 * j + l31;
      const float bn = p.bias[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mw + mi * 32 + mfma32_crow(r, hi);
        if (m < p.M) {
          const float v = acc[mi][j][r] + bn;
          p.out[(size_t)m * p.Cout + n] = RELU ? fmaxf(v, 0.f) : v;
        }
      }
    }
}

template <int WM, int WN, int TN>
static int launch_spin_conv_cfg(hipStream_t stream, const SpinConvParams& p, bool relu) {
  constexpr int BM = 64 * WM, BN = 32 * TN * WN;
  constexpr int LDS = 2 * (BM + BN) * 32 * 4;
  auto k1 = spin_conv3x3_kernel<WM, WN, TN, 1>;
  auto k0 = spin_conv3x3_kernel<WM, WN, TN, 0>;
  // per device and cheap: set on every launch (a process may drive several GPUs), return code checked
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(relu ? k1 : k0), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
    rap_set_last_hip_error((int)hipGetLastError());
    return RAP_ERR_HIP;
  }
  const unsigned grid = (unsigned)((p.M + BM - 1) / BM);
  if (relu) hipLaunchKernelGGL(k1, dim3(grid), dim3(256), LDS, stream, p);
  else hipLaunchKernelGGL(k0, dim3(grid), dim3(256), LDS, stream, p);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
int launch_spin_conv3x3(hipStream_t stream, const float* y, int ldy, int Cin, const float* Wt, const float* bias, const float* zeros, float* out,
                        int Cout, int M, bool relu) {
  if (M <= 0) return RAP_OK;
  if (Cin % 32 != 0) return RAP_ERR_INVALID;
  SpinConvParams p{y, ldy, Cin, Wt, bias, zeros, out, Cout, M};
  if (Cout == 128) return launch_spin_conv_cfg<2, 2, 2>(stream, p, relu);
  if (Cout == 64) return launch_spin_conv_cfg<4, 1, 2>(stream, p, relu);
  if (Cout == 32) return launch_spin_conv_cfg<4, 1, 1>(stream, p, relu);
  return RAP_ERR_INVALID;
}

int launch_spin_conv3d(hipStream_t stream, const float* x0, const float* Wt448, const float* bias, const float* zeros, float* out, int M) {
  if (M <= 0) return RAP_OK;
  constexpr int LDS = 2 * (256 + 64) * 32 * 4;
  auto kern = spin_conv3x3_kernel<4, 1, 2, 1, 1>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
    rap_set_last_hip_error((int)hipGetLastError());
    return RAP_ERR_HIP;
  }
  SpinConvParams p{x0, 16, 16, Wt448, bias, zeros, out, 64, M};
  hipLaunchKernelGGL(kern, dim3((unsigned)((M + 255) / 256)), dim3(256), LDS, stream, p);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

// Wt[n][tap * Cin + ci] = W[n][ci * 9 + tap] * s (BatchNorm folded), bias likewise -- the tap-major weights of spin_conv3x3_kernel
// (ntap = 9 for the 2-d layers; 27 for the Conv3d, whose rows are padded with zeros to ldt = 448)
__global__ __launch_bounds__(256) void spin_fold_tapmajor_kernel(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ rm,
                                                                 const float* __restrict__ rv, int Cin, int ntap, int ldt, float* __restrict__ Wt,
                                                                 float* __restrict__ bout) {
  const int n = blockIdx.x;
  float s = 1.f, sh = 0.f;
  if (rv) { s = 1.f / sqrtf(rv[n] + 1e-5f); sh = -rm[n] * s; }
  for (int c = threadIdx.x; c < ldt; c += 256) {
    const int tap = c / Cin, ci = c - tap * Cin;
    Wt[(size_t)n * ldt + c] = tap < ntap ? W[(size_t)n * ntap * Cin + ci * ntap + tap] * s : 0.f;
  }
  if (threadIdx.x == 0) bout[n] = b[n] * s + sh;
}
int launch_spin_fold_tapmajor(hipStream_t stream, const float* W, const float* b, const float* rm, const float* rv, int Cin, int Cout, int ntap, int ldt,
                              float* Wt, float* bout) {
  hipLaunchKernelGGL(spin_fold_tapmajor_kernel, dim3(Cout), dim3(256), 0, stream, W, b, rm, rv, Cin, ntap, ldt, Wt, bout);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
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
