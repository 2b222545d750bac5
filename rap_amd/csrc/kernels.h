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
  EPI_BIAS_RELU = 6,     // C = max(acc + bias[n], 0)      (MiniSpinNet convolutions with folded BatchNorm)
  EPI_SPLITK_PART = 7,   // internal: C[blockIdx.y][m][n] = acc over this block's share of the k-tiles (see splitk_ws)
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
  // few-row calls of the bias + residual epilogue with a long K (the FFN down-projection at a few thousand tokens): when
  // splitk_ws (>= 4 * M * N floats) is given and the tile grid covers at most half (K >= 1024) / a quarter (K >= 512) of the CUs, K is split over 4
  // blocks per tile writing partial tiles to splitk_ws, and a combine pass adds residual + bias + partials (deterministic order)
  float* splitk_ws;
  int splitk_planes;   // EPI_BIAS_SILU (round 6, the head of few-token calls): fp32 planes of M x N floats available in splitk_ws (2 or 4; 0 = no split)
  // EPI_QKV_HEADMAJOR with gamma_q / gamma_k set: MultiHeadRMSNorm (norm.py:28-33) fused -- q and k leave the GEMM normalised
  // (x / max(|x|, 1e-12) * gamma * 8 per head row), v unchanged; NULL = plain projection
  const float* gamma_q; const float* gamma_k;
  int stagger;     // set by launch_gemm_f32 (tuning key 4): 0 off, 1 first-wave blocks [256,512) start half a tile late, 2 by CU slot
  int geglu_fast;  // set by launch_gemm_f32 (tuning key 9): 1 = GEGLU on the packed fp32 pipe with the 1.5e-7 erfc (half.h geglu_pairs), 0 = erff
};
int launch_gemm_f32(hipStream_t stream, int epilogue, const GemmParams& p);

// ---------------------------------------------------------------------------------------------
// K7: variable-length, non-causal softmax attention on head-major q/k/v ([3][H][TP][64] fp32).
// Work items are (segment start, segment length, first query row) triples built on device.
// ---------------------------------------------------------------------------------------------
#define RAP_ATTN_BQ 256
struct AttnWorkItem { int seg_start, seg_len, q0, pad; };
// block_queries: query rows per work item; 0 = what the selected fp32 attention variant uses (256 or 512),
// the 16-bit attention kernel always takes 256.
// sort_ws: nseg ints of scratch -> items in longest-segment-first order (round 4); nullptr -> segment order
int launch_build_attn_worklist(hipStream_t stream, const int32_t* cu_seqlens, int nseg, AttnWorkItem* items,
                               int max_items, int block_queries, int32_t* sort_ws = nullptr);
// bound: optional per-head upper bounds (device, H floats) on the logits q.k/8 -> bounded-softmax instantiation
// splits > 1 (few-token calls, needs bound): key ranges per work item, partial O in part_o [splits][TP][heads*64] and partial row
// sums in part_l [splits][TP][heads], then one combine pass.  attention_f32_splits() picks the count for a work list (1-4).
// cover_cu / cover_nseg (splits > 1): the (sanitised) segment table the work list was built from -- the combine pass writes ZEROS for
// rows outside [cu[0], cu[nseg]), which no block produced partials for (it used to normalise stale workspace bytes there)
int launch_attention_f32(hipStream_t stream, const float* qkv_headmajor, float* out, int TP, int heads,
                         const AttnWorkItem* items, int max_items, const float* bound, float* part_o, float* part_l, int splits,
                         const int32_t* cover_cu = nullptr, int cover_nseg = 0);
int attention_f32_splits(int max_items, int heads, bool bounded);

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
                         const float* feat, int feat_dim, float* astatic, int TP, int ld = 128);
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
// out[i] = running maximum of clamp(cu[i], 0, limit): a segment table that is safe to index with, equal to cu when cu is consistent
int launch_sanitize_cu(hipStream_t stream, const int32_t* cu, int n, long limit, int32_t* out);
int launch_part_offsets(hipStream_t stream, const int64_t* points_per_part, int nparts, int32_t* part_offsets, long limit = -1);
int launch_check_batch(hipStream_t stream, const int64_t* points_per_part, const int32_t* cu_batch, int B, int P, long TP, int32_t* flag);
int launch_poison_on_flag(hipStream_t stream, const int32_t* flag, float* buf, long n);
// weight packing helpers (model creation)
int launch_copy_cols(hipStream_t stream, const float* src, int src_ld, int src_col0, float* dst, int dst_ld,
                     int dst_col0, int rows, int cols);
int launch_geglu_interleave(hipStream_t stream, const float* W, const float* b, float* Wp, float* bp, int inner,
                            int K);
int launch_fill_zero(hipStream_t stream, float* p, size_t n);

// ---------------------------------------------------------------------------------------------
// Reduced-precision (bf16 / fp16 MFMA, fp32 accumulate) twins of the transformer-block kernels.
// dtype: RAP_DT_BF16 = 1, RAP_DT_F16 = 2 (half.h).  16-bit tensors are passed as uint16_t*.
// ---------------------------------------------------------------------------------------------
enum GemmEpilogueH {
  EPI_H_BIAS = 0,            // C half (M,N) = acc (+ bias[n])
  EPI_H_BIAS_RESID_F32 = 1,  // C fp32 (M,N) = (resid +) acc (+ bias[n])      (resid may alias C)
  EPI_H_GEGLU = 3,           // W rows pre-interleaved [32 value | 32 gate]: C half (M,N/2) = (h + bh) * gelu_erf(g + bg)
  EPI_H_QKV = 4,             // N = 3*H*64: q,k -> C half [2][H][M][64]; v -> vt half [H][vt_nblk][64 d][64 pos] (half.h vt_pos)
  // 6 is retired: it meant "residual + LayerNorm" in round 2 and "fp16 residual" in round 3 under the same number;
  // the fp16-residual epilogue is 7 since ABI version 4 and 6 is refused.
  EPI_H_BIAS_RESID_H16 = 7,  // C fp16 (M,N) = fp16(resid_h (fp16) + acc + bias[n]), ONE rounding from the fp32 sum   (resid_h may alias C):
                             // the residual GEMMs of the 16-bit residual stream (rap_model_set_residual_dtype), whatever the operand dtype
  EPI_H_QKV_NORM = 5,        // EPI_H_QKV with MultiHeadRMSNorm (norm.py:28-33) fused: q,k rows are normalised from the fp32 accumulators,
                             // multiplied by gamma and by q_mul / 8 before the single rounding to 16 bit (phase-split kernel only)
};
struct GemmParamsH {
  const uint16_t* A; int lda;
  const uint16_t* W; int ldw;
  void* C; int ldc;
  int M, N, K;
  const float* bias;
  const float* resid; int ldr;
  int heads;
  uint16_t* vt; int vt_nblk;
  const uint16_t* resid_h = nullptr;                                                     // EPI_H_BIAS_RESID_H16 (row stride ldr)
  const float* gamma_q = nullptr; const float* gamma_k = nullptr; float q_mul = 8.0f;   // EPI_H_QKV_NORM
  // EPI_H_BIAS_RESID_F32 / _H16 on few-row calls (round 3): when splitk_ws (>= gemm_h16_splits(M, N, K) * M * N floats) is given and
  // gemm_h16_splits() > 1, K is split over that many blocks per 128 x 128 tile (gridDim.y) writing fp32 partial tiles to splitk_ws, and a
  // combine pass forms residual + (bias + partials) in a fixed order (deterministic; differs from the unsplit sum in fp32 rounding only)
  float* splitk_ws = nullptr;
  // round 6, few-token calls: leave the `splits` (>= 1: force_splits, else gemm_h16_splits) partial planes in splitk_ws and skip the combine
  // pass -- the caller's launch_resid_combine_ln_h16 forms the new residual-stream value AND the following LayerNorm from them
  int defer_combine = 0;
  int force_splits = 0;
  // RAP_DT_F32X2 (split precision): the weight planes are stored multiplied by a power of two (their tails stay normal fp16 numbers);
  // every epilogue multiplies the accumulators by acc_scale = its inverse (exact) first
  float acc_scale = 1.0f;
  // RAP_DT_F32X2, round 6: the 16-bit OUTPUT planes of this GEMM (V^T image of EPI_H_QKV[_NORM], GEGLU output) are stored times this
  // power of two -- chosen per tensor from the producing weights' own scale, so that a model whose weights are uniformly small (large)
  // does not push the activation's fp16 tails into the subnormals (its heads past 65 504); the consumer GEMM's acc_scale carries the inverse
  float out_scale = 1.0f;
};
// 1 = no split; 2 or 4 for GEMMs with K >= 1024 whose 128 x 128 tile grid covers at most a quarter / half of the CUs (tuning key 6)
int gemm_h16_splits(int M, int N, int K);
// the same rule without tuning key 6: what a workspace has to reserve (the key only gates the launch)
int gemm_h16_splits_by_shape(int M, int N, int K);
int launch_gemm_h16(hipStream_t stream, int dtype, int epilogue, const GemmParamsH& p);
int launch_convert_h16(hipStream_t stream, int dtype, const float* src, uint16_t* dst, size_t n);
// attention on q,k half [2][H][TP][64] + transposed-blocked v (vt); out half (TP, H*64)
// bound: optional per-head upper bounds (device, H floats) on the logits q.k/8 -- enables the bounded-softmax kernel (bf16)
// q_prescaled: q was written by launch_qknorm_h16(..., RAP_QMUL_PRESCALED) -- scores arrive in log2 units (needs bound, bf16)
// bq: the query rows per work item the list was built with (attention_h16_block_queries: 256; 64 / 128 for few-token calls, which also take
// the four-stage K / V^T ring)
int launch_attention_h16(hipStream_t stream, int dtype, const uint16_t* qk, const uint16_t* vt, int vt_nblk, uint16_t* out,
                         int TP, int heads, const AttnWorkItem* items, int max_items, const float* bound, int q_prescaled, int bq = 256,
                         int kg = 1);      // kg: key groups per block (1; 4 with bq = 64, 2 with bq = 128: attention_h16_kgroup_kernel)
bool attention_h16_wants_prescaled_q(int dtype, bool bounded);
int attention_h16_block_queries(int dtype, long rows = 0);      // work-list granularity of the 16-bit attention for a call of `rows` token rows
bool attention_h16_forced();                                    // tuning key 20 holds one of its A/B values (a forced item size / key groups)
int attention_h16_key_groups(int dtype, long rows = 0);         // key groups per block for the same call (1 unless tuning key 20 says so)
// per-head logit bounds of one attention branch after qk-norm: out[h] = 8 * max_j|gamma_q[h][j]| * max_j|gamma_k[h][j]|
int launch_qk_logit_bound(hipStream_t stream, const float* gamma_q, const float* gamma_k, int heads, float* out);
// LayerNorm with 16-bit output (fp32 statistics), same modulation forms as launch_layernorm_*
// x: the residual stream, fp32 (x_f16 = 0) or fp16 (x_f16 = 1: the 16-bit residual stream); statistics and modulation in fp32
int launch_layernorm_mod_h16(hipStream_t stream, int dtype, const void* x, int x_f16, uint16_t* out, int TP, int d, const float* mod,
                             long mod_stride, const int32_t* token_row);
int launch_layernorm_affine_h16(hipStream_t stream, int dtype, const void* x, int x_f16, uint16_t* out, int TP, int d,
                                const float* gain, const float* shift);
int launch_resid_combine_ln_h16(hipStream_t stream, int dtype, const float* part, int splits, const float* bias, void* h, int h_f16, uint16_t* out,
                                int rows, int d, const float* mod, long mod_stride, const int32_t* token_row, const float* gain,
                                const float* shift);
int launch_convert_f16_to_f32(hipStream_t stream, const uint16_t* src, float* dst, size_t n);
int launch_convert_f16_sat(hipStream_t stream, const float* src, uint16_t* dst, size_t n);    // fp32 -> fp16, saturating at +-65504
// q_mul: factor of the q plane (8 = the reference's sqrt(Dh); RAP_QMUL_PRESCALED = log2(e) for the pre-scaled attention path)
#define RAP_QMUL_PRESCALED 1.44269504088896340736f
int launch_qknorm_h16(hipStream_t stream, int dtype, uint16_t* qk, int TP, int heads, const float* gamma_q,
                      const float* gamma_k, float q_mul);

// ---------------------------------------------------------------------------------------------
// Split precision (compute dtype RAP_DT_F32X2 = 3, half.h): fp32-accurate products on the fp16 matrix pipe.  Operands are fp16 head /
// tail planes in the paired layout (a logical row of K values = 2K fp16); launch_gemm_h16 / launch_layernorm_*_h16 take dtype 3 with
// PHYSICAL K / lda / ldw (and ldc of the GEGLU output); attention has its own kernel (attn_x2.hip).
// ---------------------------------------------------------------------------------------------
int launch_x2_pack(hipStream_t stream, const float* src, long ld_src, long rows, int cols, float scale, uint16_t* dst);
int launch_x2_unpack(hipStream_t stream, const uint16_t* src, long rows, int cols, float inv_scale, float* dst);
int launch_max_abs(hipStream_t stream, const float* x, size_t n, float* out);      // *out must be 0 before; non-negative result
// q, k [2][H][2 chunks][TP][64 physical]; vt [H][vt_nblk][2 chunks][64 d][64 physical]; out paired (TP, 2 * H * 64); online softmax
// splits > 1 (few-token calls): key ranges per work item, partial O in part_o [splits][TP][heads*64] fp32, (max, row sum) in part_ml
// [splits][TP][heads][2], then one combine pass.  attention_x2_splits() picks the count for a work list (1, 2 or 4).
int launch_attention_x2(hipStream_t stream, const uint16_t* qk, const uint16_t* vt, int vt_nblk, uint16_t* out, int TP, int heads,
                        const AttnWorkItem* items, int max_items, float* part_o = nullptr, float* part_ml = nullptr, int splits = 1,
                        int n_tokens = 0,       // n_tokens: rows the combine pass covers (the real token count; 0 = TP)
                        const int32_t* cover_cu = nullptr, int cover_nseg = 0);      // as launch_attention_f32
int attention_x2_splits(int max_items, int heads);

// ---------------------------------------------------------------------------------------------
// generation selection by rigidity (rigidity.hip; reference modeling.py:456-592, eval/metrics.py:511-622)
// ---------------------------------------------------------------------------------------------
int launch_rigidity_rmse(hipStream_t stream, const float* src, const float* tgt, const float* R, const float* t,
                         const int32_t* part_offsets, int B, int P, const float* scales, int average_per_part, float* out,
                         double* partials);
int launch_step_mean(hipStream_t stream, const float* per_step, int S, int B, float* out);
int launch_select_generation(hipStream_t stream, const float* rmse, int G, int B, int P, long TP, const int32_t* cu_batch,
                             const float* clouds, const float* R, const float* t, int largest, int32_t* best, float* cloud_out,
                             float* R_out, float* t_out);

// output transforms (transforms.hip; reference eval/evaluator.py:383-490)
int launch_relative_transforms(hipStream_t stream, const float* R_pred, const float* t_pred, const float* R_gt, const float* t_gt,
                               const float* scales, const int64_t* ppp, int B, int P, const float* R_glob, const float* t_glob,
                               float* out);

// per-part rotation / translation errors relative to the anchor part (transforms.hip; reference eval/metrics.py:165-303, no-ICP branch)
int launch_transform_errors(hipStream_t stream, const float* R_gt, const float* t_gt, const float* R_pred, const float* t_pred,
                            const int64_t* ppp, const uint8_t* anchor, const int64_t* matched, const float* scale, int B, int P,
                            float* rot_pp, float* trans_pp, float* rot_mean, float* trans_mean);

// cross-part overlap ratio (overlap.hip; reference eval/metrics.py:625-691)
size_t overlap_max_items(long TP, int B);
int launch_overlap_ratio(hipStream_t stream, const float* pts, const int32_t* cu_batch, const int32_t* part_off, int B, int P,
                         long TP, const float* h_taus, int n_taus, float* ratios, float* min_dist, int32_t* pid, void* items_ws);

// MiniSpinNet local feature extractor (spinnet.hip; reference dataset_process/utils/spinnet/*)
int launch_spin_fold(hipStream_t stream, const float* W, const float* b, const float* gamma, const float* beta, const float* rm,
                     const float* rv, int Cout, int Kd, float* Wout, int ldw, int Npad, float* bout);
int launch_spin_patch(hipStream_t stream, const float* pts, const int32_t* perm, long N, const float* kpts, int K, float des_r,
                      const float* vox, const float* h_w1, const float* h_b1, float* x0, int lrf = 0);
int launch_spin_im2col3d(hipStream_t stream, const float* x0, int K, float* A, int ldA);
int launch_spin_im2col2d(hipStream_t stream, const float* y, int ldy, int Cin, int K, float* A);
int launch_spin_pool(hipStream_t stream, const float* x, int ldx, int K, const void* d_pool_w, float* desc);

// nearest-neighbour metrics (nn_metrics.hip; reference eval/metrics.py:14-48, 386-469)
struct NnWork { int x_start, x_len, q0, y_start, y_len, pad0, pad1, pad2; };
size_t nn_max_items(long n, int B);
int launch_chamfer_rmse(hipStream_t stream, const float* gt, const float* pred, const int32_t* cu_batch, int B, long TP, float* out,
                        float* d2a, float* d2b, NnWork* items);
int launch_correspondence_rmse(hipStream_t stream, const float* source_gt, const float* target_gt, const float* source_pred,
                               const float* target_pred, int Ns, int Nt, float thr, float* out3, float* d2, int32_t* nn, NnWork* items);

// farthest point sampling (fps.hip; reference dataset_process/utils/point_sampling_utils.py:263-305)
int launch_fps(hipStream_t stream, const float* pts, const int32_t* cloud_start, const int32_t* cloud_len, const int32_t* Ks,
               const int32_t* starts, int C, int Kmax,
               float* dist, int32_t* out);

// voxel down-sampling (voxel.hip; reference dataset_process/utils/dataset_utils.py:279-322)
int launch_voxel_bounds(hipStream_t stream, const float* pts, long N, float vs, long long* bounds6, unsigned int* dmax_bits);
int launch_voxel_downsample(hipStream_t stream, const float* pts, long N, float vs, const long long* h_bounds6, float dmax,
                            unsigned long long* table, long slots, unsigned int* block_cnt, unsigned int* total, long long* idx_out);

// ---------------------------------------------------------------------------------------------
// input side of the boundary (collate.hip; reference data/dataset.py:733-900 evaluation split, data/datamodule.py:169-198)
// ---------------------------------------------------------------------------------------------
size_t collate_workspace_bytes(int B, int P);
int launch_collate_transform(hipStream_t stream, const void* pts, int f64, const int64_t* points_per_part, int B, int P, long TP,
                             const int64_t* order, const float* feat_in, int F, float* cond, float* gt, float* feat_out,
                             uint8_t* anchor_idx, int64_t* part_idx, float* rotations, float* translations, float* scales,
                             uint8_t* anchor_parts, float* global_translation, int64_t* cu_seqlens, int32_t* order_flag, void* ws);

// statistical outlier removal (outlier.hip; Open3D remove_statistical_outlier as extract_sample_features.py:378-385 calls it)
size_t outlier_workspace_bytes(long N);
int launch_statistical_outliers(hipStream_t stream, const float* pts, long N, int nb_neighbors, double std_ratio, int64_t* idx_out,
                                int32_t* count_out, double* stats_out, void* ws);
int launch_voxel_coverage(hipStream_t stream, const float* pts, long N, float vs, const long long* h_bounds6, unsigned char* table, long slots,
                          unsigned long long* total);
// voxel_sort.hip: the same two results from a radix sort, O(N) memory (for grids whose dense table would be large against N)
size_t voxel_sorted_workspace_bytes(long N);
int launch_voxel_downsample_sorted(hipStream_t stream, const float* pts, long N, float vs, const long long* h_bounds6, float dmax, void* ws,
                                   size_t ws_bytes, long long* idx_out, int* count_out);
int launch_voxel_coverage_sorted(hipStream_t stream, const float* pts, long N, float vs, const long long* h_bounds6, void* ws, size_t ws_bytes,
                                 long long* count_out);
int launch_spin_conv3x3(hipStream_t stream, const float* y, int ldy, int Cin, const float* Wt, const float* bias, const float* zeros, float* out,
                        int Cout, int M, bool relu);
int launch_spin_fold_tapmajor(hipStream_t stream, const float* W, const float* b, const float* rm, const float* rv, int Cin, int Cout, int ntap, int ldt,
                              float* Wt, float* bout);
int launch_spin_conv3d(hipStream_t stream, const float* x0, const float* Wt448, const float* bias, const float* zeros, float* out, int M);
