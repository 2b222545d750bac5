// adaLN modulation table (gfx950):  for every row (a sample, or a flow step when t is uniform over
// the batch) and every adaptive LayerNorm j of the network
//     emb = Lin3_j( SiLU( Lin2_j( SiLU( Lin1_j( TS256(t_row) ) ) ) ) )      -> (scale | shift), 2d floats
// Reference: AdaptiveLayerNorm.forward (flow_model/norm.py:71-73) with diffusers
// Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0) and TimestepEmbedding (norm.py:50-55):
//     TS256(t) = [cos(t w_i), sin(t w_i)],  w_i = exp(-ln(1e4) * i / 128),  t raw in (0,1] (modeling.py:674).
// The work is tiny (weights 0.92 M floats per LN, read once): one wave per output feature, all rows
// accumulated in registers, weights streamed once with coalesced loads.
#include "kernels.h"

__global__ __launch_bounds__(256) void timestep_sinusoid_kernel(const float* __restrict__ t, int rows, float* __restrict__ ts) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= rows * 128) return;
  const int r = gid >> 7, i = gid & 127;
  // diffusers: exponent = -math.log(10000) * arange(half, float32) / half ; emb = exp(exponent) ; arg = t * emb
  const float expo = (-9.210340371976184f * (float)i) / 128.0f;
  const float w = expf(expo);
  const float arg = t[r] * w;
  ts[r * 256 + i] = cosf(arg);
  ts[r * 256 + 128 + i] = sinf(arg);
}

// Y[r][j][n] = act( sum_k X[r][xj][k] * W[j][n][k] + b[j][n] ),  xj = (x_per_ln ? j : 0)
// grid = (N/4, n_ln); wave -> output n; rows processed in chunks of 8.
template <bool SILU>
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ X, long x_row_stride, long x_ln_stride,
                                                           const float* __restrict__ W, const float* __restrict__ bias,
                                                           float* __restrict__ Y, long y_row_stride, long y_ln_stride,
                                                           int rows, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int j = blockIdx.y;
  if (n >= N) return;
  const float* w = W + ((size_t)j * N + n) * K;
  const float bn = bias[(size_t)j * N + n];
  for (int r0 = 0; r0 < rows; r0 += 8) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int k = lane; k < K; k += 64) {
      const float wk = w[k];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = r0 + i;
        if (r < rows) acc[i] += wk * X[(size_t)r * x_row_stride + (size_t)j * x_ln_stride + k];
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float s = wave_sum(acc[i]) + bn;
      const int r = r0 + i;
      if (lane == 0 && r < rows) {
        const float y = SILU ? s / (1.0f + expf(-s)) : s;
        Y[(size_t)r * y_row_stride + (size_t)j * y_ln_stride + n] = y;
      }
    }
  }
}

int launch_adaln_table(hipStream_t stream, const float* t, int rows, int n_ln, int d, const float* W1, const float* b1,
                       const float* W2, const float* b2, const float* W3, const float* b3, float* scratch, float* out) {
  if (rows <= 0 || n_ln <= 0) return RAP_OK;
  float* ts = scratch;                                   // rows * 256
  float* y1 = ts + (size_t)rows * 256;                   // rows * n_ln * d
  float* y2 = y1 + (size_t)rows * n_ln * d;              // rows * n_ln * d
  hipLaunchKernelGGL(timestep_sinusoid_kernel, dim3((rows * 128 + 255) / 256), dim3(256), 0, stream, t, rows, ts);
  RAP_LAUNCH_CHECK();
  // linear_1 (256 -> d) + SiLU   (TimestepEmbedding.act between linear_1 and linear_2)
  hipLaunchKernelGGL(small_linear_kernel<true>, dim3((d + 3) / 4, n_ln), dim3(256), 0, stream, ts, (long)256, (long)0, W1, b1,
                     y1, (long)n_ln * d, (long)d, rows, d, 256);
  RAP_LAUNCH_CHECK();
  // linear_2 (d -> d) + SiLU     (AdaptiveLayerNorm.activation before .linear, norm.py:72)
  hipLaunchKernelGGL(small_linear_kernel<true>, dim3((d + 3) / 4, n_ln), dim3(256), 0, stream, y1, (long)n_ln * d, (long)d, W2,
                     b2, y2, (long)n_ln * d, (long)d, rows, d, d);
  RAP_LAUNCH_CHECK();
  // linear (d -> 2d): scale | shift
  hipLaunchKernelGGL(small_linear_kernel<false>, dim3((2 * d + 3) / 4, n_ln), dim3(256), 0, stream, y2, (long)n_ln * d, (long)d,
                     W3, b3, out, (long)n_ln * 2 * d, (long)2 * d, rows, 2 * d, d);
  RAP_LAUNCH_CHECK();
  return RAP_OK;
}
