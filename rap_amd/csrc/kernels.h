// Internal launcher interface of librapflow (host side, C++).  Every launcher enqueues on `stream`,
// never synchronises, never allocates, and returns RAP_OK or a negative RAP_ERR_* code.
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------------------------
// K5/K8/K10/K11/K12 and the K1 projections: C = A (M,K) * W(N,K)^T with a fused epilogue.
// fp32 in, fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact fp32, bitwise an fmaf chain).
// ---------------------------------------------------------------------------------------------
enum GemmEpilogue {
  EPI_BIAS = 0,          // C = acc (+ bias[n])
  EPI_BIAS_RESID = 1,    // C = resid + acc (+ bias[n])           (resid may alias C)
  EPI_BIAS_SILU = 2,     // C = silu(acc + bias[n])
  EPI_GEGLU = 3,         // W rows pre-interleaved [32 value | 32 gate]: C(M,N/2) = (h + bh) * gelu_erf(g + bg)
  EPI_QKV_HEADMAJOR = 4, // N = 3*H*64: scatter to [3][H][M][64]
  EPI_BIAS_ANCHOR = 5,   // C = acc + bias[n] + anchor_emb[anchor[m] ? 1 : 0][n]
};

struct GemmParams {
  const float* A; int lda;
  const float* W; int ldw;
  float* C; int ldc;
  int M, N, K;
  const float* bias;
  const float* resid; int ldr;
  const uint8_t* anchor; const float* anchor_emb;
  int heads;
};
int launch_gemm_f32(hipStream_t stream, int epilogue, const GemmParams& p);

// ---------------------------------------------------------------------------------------------
// K7: variable-length, non-causal softmax attention on head-major q/k/v ([3][H][TP][64] fp32).
// Work items are (segment start, segment length, first query row) triples built on device.
// ---------------------------------------------------------------------------------------------
#define RAP_ATTN_BQ 256
struct AttnWorkItem { int seg_start, seg_len, q0, pad; };
int launch_build_attn_worklist(hipStream_t stream, const int32_t* cu_seqlens, int nseg, AttnWorkItem* items,
                               int max_items);
int launch_attention_f32(hipStream_t stream, const float* qkv_headmajor, float* out, int TP, int heads,
                         const AttnWorkItem* items, int max_items);

// ---------------------------------------------------------------------------------------------
// memory-bound ring
// ---------------------------------------------------------------------------------------------
// K4 / K9: out = LN(x) * g + b.   adaLN: g = 1 + mod[row_of(t)][0:d], b = mod[row_of(t)][d:2d]
// (mod row stride `mod_stride` floats; token_row == nullptr -> row 0 for every token).
// affine: g = gain[c], b = shift[c].
int launch_layernorm_mod(hipStream_t stream, const float* x, float* out, int TP, int d, const float* mod,
                         long mod_stride, const int32_t* token_row);
int launch_layernorm_affine(hipStream_t stream, const float* x, float* out, int TP, int d, const float* gain,
                            const float* shift);
// K6: in-place on head-major q and k.
int launch_qknorm(hipStream_t stream, float* qkv_headmajor, int TP, int heads, const float* gamma_q,
                  const float* gamma_k);
// K1 feature builders (posenc): ax (TP,64) from x_t; astatic (TP,128) from cond / scale / features.
int launch_posenc_x(hipStream_t stream, const float* x, float* ax, int TP);
int launch_posenc_static(hipStream_t stream, const float* cond, const float* scales, const int32_t* token_sample,
                         const float* feat, int feat_dim, float* astatic, int TP);
// K3: adaLN modulation table.  t (rows,), out (rows, n_ln, 2d).  scratch: rows*n_ln*(256/n_ln + 2d) floats.
int launch_adaln_table(hipStream_t stream, const float* t, int rows, int n_ln, int d, const float* W1,
                       const float* b1, const float* W2, const float* b2, const float* W3, const float* b3,
                       float* scratch, float* out);
// K12 tail: v (TP,3) = y (TP,K) * W(3,K)^T
int launch_head_out3(hipStream_t stream, const float* y, int ldy, const float* W, float* v, int TP, int K);
// K13: x0hat = x - v*t ; x_new = x - dt*v   (two roundings each, as the reference's tensor ops).
int launch_euler_step(hipStream_t stream, const float* x_t, const float* v, float t, float dt, float* x0hat_out,
                      float* x_next_out, float* traj_xt_slot_or_null, long n);
// K14-K16
#define RAP_PROC_CHUNKS 16
int launch_procrustes_fit(hipStream_t stream, const float* src, const float* tgt, const int32_t* part_offsets,
                          int nparts, float* R_out, float* t_out, double* partials);
int launch_rigid_apply(hipStream_t stream, const float* src, const float* R, const float* t,
                       const int32_t* part_offsets, int nparts, float* out, const float* x1, float w0, float w1,
                       float* traj_slot_or_null, int blend);
// segment tables
int launch_token_sample(hipStream_t stream, const int32_t* cu_batch, int B, int32_t* token_sample);
int launch_part_offsets(hipStream_t stream, const int64_t* points_per_part, int nparts, int32_t* part_offsets);
// weight packing helpers (model creation)
int launch_copy_cols(hipStream_t stream, const float* src, int src_ld, int src_col0, float* dst, int dst_ld,
                     int dst_col0, int rows, int cols);
int launch_geglu_interleave(hipStream_t stream, const float* W, const float* b, float* Wp, float* bp, int inner,
                            int K);
int launch_fill_zero(hipStream_t stream, float* p, size_t n);
