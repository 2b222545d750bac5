// Positional-encoding feature builders for the embedding projection (gfx950).
//
// Reference: PointCloudEmbedding.embed (flow_model/embedding.py:29-58) and
// PointCloudEncodingManager.forward (embedding.py:131-179):
//     emb_in = cat[ PE63(cond), PE63(x_t), PE21(scale), feat ]  ->  emb_proj (179 -> 512)
//     PE(v)  = [ v, sin(2^0 v), cos(2^0 v), ..., sin(2^9 v), cos(2^9 v) ]   (component-major inside each block)
//
// Only the 63 x_t columns change between flow steps, so the projection is split:
//     base  = [PE63(cond), PE21(scale), feat, 0-pad] (TP,128) * Wstatic^T + bias + anchor_emb     (once per sample call)
//     embed = [PE63(x_t), 0] (TP,64) * Wx^T + base                                                  (every step)
// These kernels write the two feature matrices; the GEMM kernel does the projections.
// sin/cos use the full-range-reduction sinf/cosf (arguments reach 512*|x| ~ 2500 rad; the fast
// __sinf path would break parity).  2^k * v is exact in fp32, so the argument equals the reference's.
#include "kernels.h"

// 16 threads per token: thread j < 10 -> frequency j (6 outputs); j == 10 -> raw xyz + pad column 63.
__global__ __launch_bounds__(256) void posenc_x_kernel(const float* __restrict__ x, float* __restrict__ ax, int TP) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long tok = gid >> 4;
  const int j = (int)(gid & 15);
  if (tok >= TP) return;
  const float* xp = x + tok * 3;
  float* o = ax + tok * 64;
  if (j < 10) {
    const float f = (float)(1 << j);
    const float a0 = xp[0] * f, a1 = xp[1] * f, a2 = xp[2] * f;
    float* q = o + 3 + 6 * j;
    q[0] = sinf(a0); q[1] = sinf(a1); q[2] = sinf(a2);
    q[3] = cosf(a0); q[4] = cosf(a1); q[5] = cosf(a2);
  } else if (j == 10) {
    o[0] = xp[0]; o[1] = xp[1]; o[2] = xp[2];
    o[63] = 0.f;
  }
}

int launch_posenc_x(hipStream_t stream, const float* x, float* ax, int TP) {
  if (TP <= 0) return RAP_OK;
  const long nthreads = (long)TP * 16;
  hipLaunchKernelGGL(posenc_x_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, stream, x, ax, TP);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}

// 32 threads per token.  Output row (128 floats):
//   [0,63)    PE63(cond)          [63,84)  PE21(scale of the token's sample)
//   [84,84+F) local features      rest     zero padding            (F <= 44, multiple of 4)
__global__ __launch_bounds__(256) void posenc_static_kernel(const float* __restrict__ cond, const float* __restrict__ scales,
                                                            const int32_t* __restrict__ token_sample,
                                                            const float* __restrict__ feat, int F, float* __restrict__ as, int TP, int ld) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long tok = gid >> 5;
  const int j = (int)(gid & 31);
  if (tok >= TP) return;
  float* o = as + tok * ld;                  // ld = 128, or 128 + the latent-feature columns a model with in_dim > 0 appends
  if (j < 10) {
    const float* xp = cond + tok * 3;
    const float f = (float)(1 << j);
    const float a0 = xp[0] * f, a1 = xp[1] * f, a2 = xp[2] * f;
    float* q = o + 3 + 6 * j;
    q[0] = sinf(a0); q[1] = sinf(a1); q[2] = sinf(a2);
    q[3] = cosf(a0); q[4] = cosf(a1); q[5] = cosf(a2);
  } else if (j == 10) {
    const float* xp = cond + tok * 3;
    o[0] = xp[0]; o[1] = xp[1]; o[2] = xp[2];
  } else if (j < 21) {
    const int k = j - 11;
    const float s = scales[token_sample[tok]] * (float)(1 << k);
    o[63 + 1 + 2 * k] = sinf(s);
    o[63 + 2 + 2 * k] = cosf(s);
  } else if (j == 21) {
    o[63] = scales[token_sample[tok]];
  } else {
    // threads 22..31: features and zero padding, 4 floats each starting at column 84 + 4*(j-22)
    const int c = 84 + 4 * (j - 22);
    if (c < 128) {
      const int fc = c - 84;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (fc < F) v = *reinterpret_cast<const float4*>(feat + tok * F + fc);
      *reinterpret_cast<float4*>(o + c) = v;
    }
    if (j == 31) {  // columns 124..127 (the 11th float4 group) are not covered by threads 22..31's first slot
      *reinterpret_cast<float4*>(o + 124) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

int launch_posenc_static(hipStream_t stream, const float* cond, const float* scales, const int32_t* token_sample,
                         const float* feat, int feat_dim, float* astatic, int TP, int ld) {
  if (TP <= 0) return RAP_OK;
  if (feat_dim % 4 != 0 || feat_dim > 40 || feat_dim < 0 || ld < 128 || ld % 4 != 0) return RAP_ERR_INVALID;
  const long nthreads = (long)TP * 32;
  hipLaunchKernelGGL(posenc_static_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, stream, cond, scales,
                     token_sample, feat, feat_dim, astatic, TP, ld);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
