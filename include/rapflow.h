/* librapflow -- C ABI of the MI355X-native rectified-flow registration sampler.
 *
 * Drop-in boundary for ONE hot path of PRBonn/RAP: rectified_point_flow/{sampler.py, flow_model/,
 * procrustes.py} behind RectifiedPointFlow.sample_rectified_flow (reference modeling.py:632-741).
 * The reference has no native code; its "FFI" for this path is the set of Python call sites listed
 * next to each entry point below.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer (HIP, gfx950) unless named h_*;
 *   - fp32 tensors are dense row-major; packed varlen layout (TP,C) with segment tables, exactly the
 *     reference's collate schema (data/datamodule.py:169-198);
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, the call never synchronises and never
 *     allocates (except rap_model_create / rap_model_destroy);
 *   - returns 0 (RAP_OK) or a negative code: -1 invalid argument, -2 workspace too small,
 *     -3 HIP runtime error (see rap_last_hip_error), -4 allocation failure;
 *   - the caller owns every buffer, including the workspace (size from rap_workspace_bytes).
 */
#ifndef RAPFLOW_H
#define RAPFLOW_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version of this header.  rap_version() returns the value the LIBRARY was built with; a caller compiled against another
 * value must not use the library (round 4, version 4: rap_spinnet_describe gained `flags` and three entry points were removed in
 * round 3 without a bump; the fp16-residual epilogue of rap_gemm_h16 moved from 6 to 7 and 6 is refused; rap_poison_on_flag is new;
 * round 5, version 5: compute dtype 3 (split precision) and the rap_x2_* entry points are new, nothing was removed or re-numbered;
 * round 6, version 6: additive -- the *_latent entry points (in_dim > 0), rap_transform_errors, tuning keys 18 / 19 / 20; the scratch of the
 * kernel-level attention entry points grew by a sanitised copy of cu_seqlens (rap_attention_workspace_bytes reports it)). */
#define RAPFLOW_ABI_VERSION 6

/* return codes of every int-returning entry point */
#define RAP_OK 0
#define RAP_ERR_INVALID (-1)
#define RAP_ERR_WORKSPACE (-2)
#define RAP_ERR_HIP (-3)
#define RAP_ERR_ALLOC (-4)

typedef struct rap_model rap_model;

/* PointCloudDiT hyper-parameters (reference config/model/flow_model/point_cloud_dit_12.yaml,
 * flow_model/point_cloud_dit.py:20-36).  head_dim is fixed at 64 (embed_dim == 64 * num_heads),
 * embed_dim a multiple of 256, local_feat_dim a multiple of 4 and <= 40, in_dim == 0.  The native embedding layout is the one of
 * scale_emb_on == local_feat_concat_on == true (config/RAP_inference.yaml:65); a model built with either switch off maps onto it with
 * zero weight columns / unit gains (rap_amd/flow_model.py _native_tensors), qk_norm == false through rap_model_set_qk_norm. */
typedef struct rap_model_desc {
  int32_t embed_dim;      /* 512 */
  int32_t num_layers;     /* 10 / 12 / 16 */
  int32_t num_heads;      /* 8 */
  int32_t local_feat_dim; /* 32 */
} rap_model_desc;

int rap_version(void);
int rap_last_hip_error(void);

/* Number of fp32 values in the raw weight blob: the tensors of PointCloudDiT.state_dict() concatenated
 * in the reference's registration order (flow_model/point_cloud_dit.py:83-117, layer.py:71-89,
 * norm.py:47-58; table in DESIGN.md "Weight contract").  Returns < 0 for an unsupported desc. */
int64_t rap_weight_count(const rap_model_desc* desc);

/* Replaces  PointCloudDiT(...).load_state_dict(ckpt)  (reference sample.py:58, utils/checkpoint.py:13-61).
 * Copies the blob into model-owned device memory and builds the kernel-side packings (split embedding
 * projection, stacked adaLN weights, value/gate-interleaved GEGLU projection).  Allocates. */
int rap_model_create(const rap_model_desc* desc, const float* d_weights, int64_t n_floats, void* stream,
                     rap_model** out);
/* in_dim > 0 (round 6; reference flow_model/embedding.py:107-118,163-166, point_cloud_dit.py:56,86): the model concatenates `in_dim`
 * LATENT point features (the PTv3 encoder output of encoder_on = true; modeling.py:788 builds one with in_dim = 64) into the embedding
 * input.  They are step-invariant like the condition cloud, so they become in_dim more columns of the hoisted embedding GEMM.  The blob's
 * emb_proj.weight is (embed_dim, 147 + local_feat_dim + in_dim) with the latent columns LAST ([cond 63 | x_t 63 | scale 21 | feat F |
 * latent in_dim]; rap_amd/flow_model.py re-orders the reference's [cond | x_t | latent | scale | feat]).  in_dim a multiple of 4, <= 512.
 * in_dim = 0 is rap_weight_count / rap_model_create.  A model with in_dim > 0 must be driven through the *_latent entry points below. */
int64_t rap_weight_count_latent(const rap_model_desc* desc, int32_t in_dim);
int rap_model_create_latent(const rap_model_desc* desc, int32_t in_dim, const float* d_weights, int64_t n_floats, void* stream,
                            rap_model** out);
void rap_model_destroy(rap_model* m);

/* Arithmetic type of the transformer blocks (qkv / out / feed-forward GEMMs and attention) for subsequent calls on `m`:
 * 0 = fp32 (default; exact-fp32 MFMA, the parity configuration of BASELINE configs[1]),
 * 1 = bf16 MFMA (BASELINE configs[2], [4]), 2 = fp16 MFMA,
 * 3 = SPLIT PRECISION ("float32x2", round 5): fp32-ACCURATE results from the fp16 matrix pipe.  Every operand of the block GEMMs and of
 *     both attention products is carried as an fp16 head + an fp16 tail (x = hi + lo, 22 significand bits; weights additionally scaled by
 *     a per-tensor power of two so that their tails are normal numbers) and each contraction keeps hi*hi + hi*lo + lo*hi in the MFMA's
 *     fp32 accumulators -- 3 x 32 matrix-pipe cycles per 16 contraction steps instead of 8 x 64 for the fp32-input MFMA.  Residual stream,
 *     LayerNorm, qk-norm, softmax state, GEGLU, embedding, head, Euler and Procrustes are fp32 exactly as in mode 0; results sit at
 *     mode 0's distance from an fp64 evaluation.  Operands are clipped to the fp16 range (|x| <= 65504), the range
 *     of the reference's own shipped fp16 inference.  The residual-dtype switch below is ignored in this mode.
 * Modes 1 / 2 replace the autocast context the reference runs its
 * GPU inference under (Lightning precision "16-mixed", config/trainer/infer.yaml:6; attn_dtype, layer.py:106-128).
 * Residual stream, LayerNorm, softmax, accumulation, embedding, final_mlp, Euler and Procrustes stay fp32
 * (final_mlp is fp32 in the reference too, point_cloud_dit.py:183-184).  The first call per dtype allocates and
 * fills the 16-bit weight copies.  rap_workspace_bytes depends on the current dtype. */
int rap_model_set_compute_dtype(rap_model* m, int32_t dtype, void* stream);
int rap_model_compute_dtype(const rap_model* m);
/* How many of the model's 2 * num_layers attention launches (layer x {per part, per sample}) take the bounded, offset-free softmax
 * kernel: a launch does when every head of THAT attention has 8 max|gamma_q| max|gamma_k| <= 40 (MultiHeadRMSNorm gains,
 * flow_model/norm.py:15-33); the others take the online-softmax kernel.  Decided per launch, so one hot head of a trained checkpoint
 * costs one (layer, branch) the faster kernel, not the model.  Returns < 0 for a NULL model. */
int rap_model_bounded_attention_launches(const rap_model* m);
/* qk_norm of the reference's constructor (flow_model/point_cloud_dit.py:28; layer.py:75-83,103-104): 1 (default, every shipped configuration) =
 * MultiHeadRMSNorm on q and k; 0 = q and k go to the attention as projected -- no bound on the logits exists then, every attention launch
 * takes the online-softmax kernel, and the gamma entries of the weight blob are ignored.  (The reference's other two constructor switches,
 * scale_emb_on / local_feat_concat_on, only narrow the embedding projection's input: a caller widens emb_proj.weight with zero columns for
 * the absent inputs -- rap_amd.PointCloudDiT does -- which is exact.) */
int rap_model_set_qk_norm(rap_model* m, int32_t on);
int rap_model_qk_norm(const rap_model* m);
/* Storage type of the residual stream in the 16-bit compute modes: 0 = fp32 (default), 2 = fp16.  With fp16 the stream between the
 * layer kernels is held the way the reference's own "16-mixed" inference holds it (nn.Linear outputs are 16-bit under autocast and
 * flow_model/layer.py:155-164 adds them): every residual sum is formed in fp32 from the GEMM's fp32 accumulators and rounded once,
 * LayerNorm statistics stay fp32, the head (fp32 in the reference too, point_cloud_dit.py:183-184) reads an fp32 image of it.
 * Halves the HBM bytes of the two N = 512 GEMMs and the three LayerNorms of a layer.  Ignored while the compute dtype is fp32.
 * rap_workspace_bytes depends on it.  Not a per-call switch: set it before the first call like the compute dtype. */
int rap_model_set_residual_dtype(rap_model* m, int32_t dtype);
int rap_model_residual_dtype(const rap_model* m);

/* Bytes of caller-provided workspace for one call on a batch of TP points, B samples, `nseg_part`
 * part segments (B*P for rap_sample, VP for rap_dit_forward) and `rows` adaLN rows
 * (num_steps for rap_sample, B for rap_dit_forward).  Depends on the model's compute / residual dtype (query it in the state the call
 * will run in; the setters and the enqueueing entry points serialise on a per-model mutex) and on nothing else: token-row buffers are
 * carved at align_up(TP, 256) rows (the layer kernels run over the padded rows, so any TP takes the persistent 256-row-tile GEMMs),
 * and the split-K planes of few-token 16-bit calls are reserved by shape, whatever tuning key 6 says.  A call whose workspace is too
 * small returns -2, it never overruns. */
size_t rap_workspace_bytes(const rap_model* m, int64_t TP, int32_t B, int32_t nseg_part, int32_t rows);

/* Replaces PointCloudDiT.forward (reference flow_model/point_cloud_dit.py:141-191), called from
 * modeling.py:684-706.  x_t (TP,3), timesteps (B,) raw t in (0,1], cond (TP,3), feat (TP,F), scales (B,),
 * anchor (TP,) uint8/bool, cu_batch (B+1,) int32, cu_part (VP+1,) int32 (zero-length segments allowed).
 * v_out (TP,3); feats_out (TP,embed_dim) or NULL ('transformer_features', :186-190). */
int rap_dit_forward(const rap_model* m, const float* x_t, const float* timesteps, const float* cond, const float* feat,
                    const float* scales, const uint8_t* anchor, const int32_t* cu_batch, const int32_t* cu_part,
                    int32_t B, int32_t VP, int64_t TP, float* v_out, float* feats_out, void* ws, size_t ws_bytes,
                    void* stream);
/* ... with latent (TP, in_dim) fp32 for a model created with in_dim > 0 (`latent_features` of point_cloud_dit.py:146, embedding.py:163-166);
 * latent must be NULL exactly when the model's in_dim is 0. */
int rap_dit_forward_latent(const rap_model* m, const float* x_t, const float* timesteps, const float* cond, const float* feat,
                           const float* latent, const float* scales, const uint8_t* anchor, const int32_t* cu_batch, const int32_t* cu_part,
                           int32_t B, int32_t VP, int64_t TP, float* v_out, float* feats_out, void* ws, size_t ws_bytes,
                           void* stream);

/* Replaces euler_step's tensor update (reference sampler.py:88-90): x0_hat = x_t - v*t ; x_next = x_t - dt*v.
 * n = number of floats (3*TP).  x_next may alias x_t.  traj_xt_slot may be NULL. */
int rap_euler_step(const float* x_t, const float* v, float t, float dt, float* x0_hat, float* x_next,
                   float* traj_xt_slot, int64_t n, void* stream);

/* Replaces fit_transformations (reference procrustes.py:40-84).  points_per_part (B,P) int64 as in the
 * reference; src/tgt (TP,3) packed in (b,p) order.  R_out (B,P,3,3), t_out (B,P,3); empty parts -> zero rows.
 * Convention: tgt ~= src @ R^T + t.  ws >= rap_procrustes_workspace_bytes(B*P). */
size_t rap_procrustes_workspace_bytes(int32_t nparts);
int rap_fit_transformations(const float* src, const float* tgt, const int64_t* points_per_part, int32_t B, int32_t P,
                            float* R_out, float* t_out, void* ws, size_t ws_bytes, void* stream);

/* Replaces rigidify_prediction_with_procrustes (reference procrustes.py:86-118): out = cond_p R_p^T + t_p. */
int rap_rigidify(const float* prediction, const float* condition, const int64_t* points_per_part, int32_t B, int32_t P,
                 float* out, void* ws, size_t ws_bytes, void* stream);

/* The sampler's rigidity-forcing update (reference sampler.py:59-60):
 *   x_t = rigidify(x0_hat, cond) * w0 + x_1 * w1   with w0 = 1 - t + dt, w1 = t - dt (computed in double by the caller). */
int rap_rigidify_blend(const float* x0_hat, const float* condition, const int64_t* points_per_part, int32_t B, int32_t P,
                       const float* x_1, float w0, float w1, float* x_t_out, void* ws, size_t ws_bytes, void* stream);

/* Replaces the whole of sample_rectified_flow + the final fit_transformations
 * (reference modeling.py:632-741 -> sampler.py:11-74 [euler] -> modeling.py:389-391):
 *   dt = 1/num_steps; x_t = x_1; for s: t = 1 - s*dt; v = DiT(x_t, t); x0 = x_t - v t; x_t -= dt v;
 *   if rigidity: x_t = rigid(x0, cond) (1-t+dt) + x_1 (t-dt);  traj_x0[s] = x0 (raw);  traj_xt[s] = x_t;
 *   R,t = fit_transformations(cond, traj_x0[S-1]).
 * traj_x0, traj_xt (num_steps,TP,3); R_out (B,P,3,3); t_out (B,P,3); feats_out (TP,embed_dim) or NULL
 * (transformer features of the last step, modeling.py:678-695). */
int rap_sample(const rap_model* m, const float* cond, const float* feat, const float* scales, const uint8_t* anchor,
               const int64_t* points_per_part, const int32_t* cu_batch, const float* x_1, int32_t B, int32_t P,
               int64_t TP, int32_t num_steps, int32_t rigidity_forcing, float* traj_x0, float* traj_xt, float* R_out,
               float* t_out, float* feats_out, void* ws, size_t ws_bytes, void* stream);
/* ... with the `latent_features` argument of sample_rectified_flow (modeling.py:636): (TP, in_dim) fp32, NULL exactly when in_dim is 0 */
int rap_sample_latent(const rap_model* m, const float* cond, const float* feat, const float* latent, const float* scales,
                      const uint8_t* anchor, const int64_t* points_per_part, const int32_t* cu_batch, const float* x_1, int32_t B,
                      int32_t P, int64_t TP, int32_t num_steps, int32_t rigidity_forcing, float* traj_x0, float* traj_xt, float* R_out,
                      float* t_out, float* feats_out, void* ws, size_t ws_bytes, void* stream);

/* ---- generation selection by rigidity (the caller side of the path) ----
 * Replaces compute_rigidity_rmse (reference eval/metrics.py:511-622): per sample, the RMS of |cond_p R_p^T + t_p - pred_p|
 * over the points of all non-empty parts (average_per_part: mean over parts of the per-part RMS), inf for a sample
 * without points, times scales[b] if scales != NULL.  out (B,).  ws >= rap_rigidity_workspace_bytes(B*P, steps, B). */
size_t rap_rigidity_workspace_bytes(int32_t nparts, int32_t steps, int32_t B);
int rap_rigidity_rmse(const float* cond, const float* pred, const float* R, const float* t, const int64_t* points_per_part,
                      int32_t B, int32_t P, const float* scales, int32_t average_per_part, float* out, void* ws,
                      size_t ws_bytes, void* stream);
/* Replaces the use_average_rigidity_rmse loop of test_step (reference modeling.py:466-500): for every step of the
 * end-point trajectory traj (steps,TP,3): R,t = fit_transformations(cond, traj[s]); rmse[s] = rigidity RMSE; mean_out (B,) =
 * mean over steps.  per_step_out (steps,B) or NULL. */
int rap_trajectory_rigidity_rmse(const float* cond, const float* traj, const int64_t* points_per_part, int32_t B, int32_t P,
                                 int64_t TP, int32_t steps, const float* scales, float* mean_out, float* per_step_out, void* ws,
                                 size_t ws_bytes, void* stream);
/* Replaces the rigidity-selected generation pick (reference modeling.py:518, 560-592): best_out[b] = argmin_g rmse[g][b]
 * (first minimum), or with pick_largest != 0 the overlap-ratio pick argmax_g (modeling.py:601); if clouds != NULL also gathers cloud_out (TP,3), R_out (B,P,3,3), t_out (B,P,3) of the picked generation
 * per sample from clouds (G,TP,3), R (G,B,P,3,3), t (G,B,P,3); cu_batch (B+1,) int32. */
int rap_select_generation(const float* rmse, int32_t G, int32_t B, int32_t P, int64_t TP, const int32_t* cu_batch,
                          const float* clouds, const float* R, const float* t, int32_t pick_largest, int32_t* best_out,
                          float* cloud_out, float* R_out, float* t_out, void* stream);
/* ---- registration errors (the RRE / RTE that BASELINE.json's "SE(3) err" is defined by) ----
 * Replaces compute_transform_errors(..., use_icp=False) (reference eval/metrics.py:165-303): per sample, every non-anchor, non-empty part's
 * ground-truth and predicted pose relative to the (first) anchor part's, delta_R = R_gt_rel^T R_pred_rel, delta_t = (t_pred_rel -
 * t_gt_rel) * scale[b];  rot_err = deg(acos(clamp((tr delta_R - 1) / 2, -1, 1))), trans_err = |delta_t|.  R_* (B,P,3,3), t_* (B,P,3),
 * points_per_part (B,P) int64, anchor_part (B,P) uint8, matched_part_ids (B,P) int64 or NULL (re-orders the PREDICTED poses, :223-227),
 * scale (B,) or NULL (= 1).  Outputs: per-part errors (B,P) (0 for anchor / empty parts) and their means over the valid parts (B,)
 * (NaN for a sample without one, as the reference's 0 / 0).  No workspace, no synchronisation. */
int rap_transform_errors(const float* R_gt, const float* t_gt, const float* R_pred, const float* t_pred, const int64_t* points_per_part,
                         const uint8_t* anchor_part, const int64_t* matched_part_ids, const float* scale, int32_t B, int32_t P,
                         float* rot_err_per_part, float* trans_err_per_part, float* rot_err_mean, float* trans_err_mean, void* stream);

/* Replaces compute_overlap_ratio (reference eval/metrics.py:625-691): per sample the fraction of points that have a point of
 * a DIFFERENT part of the same sample within distance tau, for n_taus (<= 8) thresholds given as a HOST array.
 * ratios_out (n_taus, B) device; min_dist_out (TP,) device or NULL (distance to the nearest other-part point, inf if none).
 * 0 for samples with <= 1 point or fewer than two non-empty parts.  ws >= rap_overlap_workspace_bytes(TP, B, P). */
size_t rap_overlap_workspace_bytes(int64_t TP, int32_t B, int32_t P);
int rap_overlap_ratio(const float* pointclouds_pred, const int64_t* points_per_part, const int32_t* cu_batch, int32_t B, int32_t P,
                      int64_t TP, const float* h_taus, int32_t n_taus, float* ratios_out, float* min_dist_out, void* ws,
                      size_t ws_bytes, void* stream);

/* ---- output transforms (the data format after the path) ----
 * Replaces the 4x4 computation of Evaluator._save_transformation_files (reference eval/evaluator.py:383-490): per (sample,
 * part) the predicted pose relative to the ground-truth pose in metres, R_rel = R_pred R_gt^T, t_rel = s t_pred -
 * (s t_gt) R_rel^T, times inv([R_global | t_global]) when the per-sample global frame (B,3,3)/(B,3) is given.
 * out (B,P,4,4) row-major fp32; all-zero blocks for parts without points. */
int rap_relative_transforms(const float* R_pred, const float* t_pred, const float* R_gt, const float* t_gt, const float* scales,
                            const int64_t* points_per_part, int32_t B, int32_t P, const float* global_rotation,
                            const float* global_translation, float* out, void* stream);

/* Nearest-neighbour metrics of the same evaluator.  ws >= rap_nn_metrics_workspace_bytes(points, B).
 * rap_chamfer_rmse replaces compute_cd (reference eval/metrics.py:14-48): per object sqrt(0.5 (mean_i min_j |gt_i - pred_j|^2 +
 * mean_j min_i |pred_j - gt_i|^2)); gt, pred (TP,3) packed by cu_batch (B+1,) int32; out (B,).
 * rap_correspondence_rmse replaces compute_correspondence_rmse (:386-469) for ONE scan pair: nearest target_gt point of every
 * source_gt point; pairs within distance_threshold are correspondences; out3 = {RMS of |source_pred_i - target_pred_nn(i)| over
 * them (inf if none), their number, number / n_source} (device, 3 floats). */
size_t rap_nn_metrics_workspace_bytes(int64_t n_points, int32_t B);
int rap_chamfer_rmse(const float* pointclouds_gt, const float* pointclouds_pred, const int32_t* cu_batch, int32_t B, int64_t TP,
                     float* out, void* ws, size_t ws_bytes, void* stream);
int rap_correspondence_rmse(const float* source_gt, const float* target_gt, const float* source_pred, const float* target_pred,
                            int32_t n_source, int32_t n_target, float distance_threshold, float* out3, void* ws, size_t ws_bytes,
                            void* stream);

/* ---- MiniSpinNet local feature extractor (the step before the path) ----
 * Replaces MiniSpinNet.forward (reference dataset_process/utils/spinnet/patch_embedder.py:49-183 with patchnet.py:16-84 and
 * utils/common.py) as extract_sample_features.py:151-220 / demo.py:959-988 call it: global-z alignment, 512 points per patch,
 * 3 x 7 x 20 voxels x 10 samples, delta 0.8 -- the only configuration the reference ships.  Weight blob = the float tensors
 * of MiniSpinNet.state_dict() in registration order (pnt_layer, pool_layer, conv_net; num_batches_tracked skipped), eval mode.
 * rap_spinnet_describe: pts (N,3) raw cloud, perm (N,) int32 or NULL = the shuffle select_patches applies before the ball
 * query (patch_embedder.py:99-100; the ball query keeps the first 512 in-radius points IN THAT ORDER), kpts (K,3), des_r;
 * desc_out (K,32) unit-norm descriptors = the `features` of the sampling path.  Keypoints are processed in chunks of
 * keypoints_per_chunk; ws >= rap_spinnet_workspace_bytes(keypoints_per_chunk) (about 0.8 MB per keypoint). */
typedef struct rap_spinnet rap_spinnet;
int64_t rap_spinnet_weight_count(void);
int rap_spinnet_create(const float* d_weights, int64_t n_floats, void* stream, rap_spinnet** out);
void rap_spinnet_destroy(rap_spinnet* m);
size_t rap_spinnet_workspace_bytes(int32_t keypoints_per_chunk);
/* flags of rap_spinnet_describe (per call: the handle is immutable after rap_spinnet_create, two callers may share it):
 *   RAP_SPINNET_PATCH_LRF   patch alignment of MiniSpinNet.forward (patch_embedder.py:141-166): unset (default) = is_aligned_to_global_z,
 *                           the shipped demo setting (R = I); set = every patch is rotated so that its own normal -- the singular vector of
 *                           the smallest singular value of the patch covariance, oriented towards the origin (cal_Z_axis,
 *                           utils/common.py:539-557) -- becomes +z (RodsRotatFormula, :472-496);
 *   RAP_SPINNET_IM2COL_PATH A/B: unset (default) = the seven 3x3 cylindrical convolutions as implicit GEMMs (spin_conv3x3_kernel: the
 *                           im2col gather happens in the LDS-DMA source addresses); set = the round-1 path (materialised im2col + the
 *                           fp32 GEMM at 128 padded columns). */
#define RAP_SPINNET_PATCH_LRF 1
#define RAP_SPINNET_IM2COL_PATH 2
int rap_spinnet_describe(const rap_spinnet* m, const float* pts, const int32_t* perm, int64_t N, const float* kpts, int32_t K,
                         float des_r, int32_t flags, float* desc_out, int32_t keypoints_per_chunk, void* ws, size_t ws_bytes, void* stream);

/* Farthest point sampling, the keypoint selection in front of MiniSpinNet (reference dataset_process/utils/
 * point_sampling_utils.py:263-305 -> pytorch3d.ops.sample_farthest_points with lengths, per-cloud K and a random start):
 * points (T,3) holds the clouds: cloud c = rows [cloud_start[c], cloud_start[c] + cloud_len[c]) (packed or zero-padded batches
 * alike); k_per_cloud / start_idx (n_clouds,) int32 (the caller draws the start indices, as pytorch3d does with torch.randint);
 * indices_out (n_clouds, k_max) int32, cloud-local, -1 padded past min(K, length); dist_ws: T floats of scratch.  Ties go to the
 * lowest index (torch.argmax). */
int rap_farthest_point_sampling(const float* points, const int32_t* cloud_start, const int32_t* cloud_len, const int32_t* k_per_cloud,
                                const int32_t* start_idx, int32_t n_clouds, int32_t k_max, int32_t* indices_out, float* dist_ws,
                                void* stream);

/* Voxel down-sampling, the first preprocessing step of the same script (reference dataset_process/utils/dataset_utils.py:279-322,
 * voxel_down_sample_torch): per occupied voxel the point closest to the voxel centre (distance quantised to 1000 levels, ties to the
 * lowest index), indices returned in ascending voxel-key order.  Two steps because the dense key table is sized from the data:
 *   rap_voxel_bounds    -> bounds6_out (device, 6 int64: per-axis min then max of floor(p / voxel_size)), dist_max_out (device float);
 *   the caller copies both to the host (the reference synchronises at the same point), then
 *   rap_voxel_downsample(points, N, voxel_size, h_bounds6 (HOST), dist_max, indices_out (device, >= min(N, slots) int64),
 *                        count_out (device int32: number of kept points), ws >= rap_voxel_workspace_bytes(h_bounds6)).
 * rap_voxel_table_slots returns the table size (8 bytes per slot), or -1 when the grid needs more than 2^33 slots. */
int rap_voxel_bounds(const float* points, int64_t N, float voxel_size, int64_t* bounds6_out, float* dist_max_out, void* stream);
int64_t rap_voxel_table_slots(const int64_t* h_bounds6);
size_t rap_voxel_workspace_bytes(const int64_t* h_bounds6);
/* Exact number of occupied voxels = distinct rows of floor(points / voxel_size) (calculate_voxel_coverage,
 * dataset_process/utils/point_sampling_utils.py:11-31; the down-sampling table above keeps the reference's colliding key and cannot
 * count).  h_bounds6 from rap_voxel_bounds (HOST); count_out: device int64; ws >= rap_voxel_coverage_workspace_bytes(h_bounds6). */
size_t rap_voxel_coverage_workspace_bytes(const int64_t* h_bounds6);
int rap_voxel_coverage(const float* points, int64_t N, float voxel_size, const int64_t* h_bounds6, int64_t* count_out, void* ws,
                       size_t ws_bytes, void* stream);
int rap_voxel_downsample(const float* points, int64_t N, float voxel_size, const int64_t* h_bounds6, float dist_max,
                         int64_t* indices_out, int32_t* count_out, void* ws, size_t ws_bytes, void* stream);
/* The same two results with O(N) memory (a radix sort of one 64-bit key per point instead of a table over the grid volume): for grids
 * whose dense table would be large against N or exceed its slot limit.  Identical outputs (kept indices in ascending voxel-key order;
 * the exact occupied-voxel count).  Limits: N < 2^32; down-sampling needs the largest grid extent v < 2^18 cells.
 * ws >= rap_voxel_sorted_workspace_bytes(N) for both.  Reference: dataset_utils.py:279-322, point_sampling_utils.py:11-31. */
size_t rap_voxel_sorted_workspace_bytes(int64_t N);
int rap_voxel_downsample_sorted(const float* points, int64_t N, float voxel_size, const int64_t* h_bounds6, float dist_max,
                                int64_t* indices_out, int32_t* count_out, void* ws, size_t ws_bytes, void* stream);
int rap_voxel_coverage_sorted(const float* points, int64_t N, float voxel_size, const int64_t* h_bounds6, int64_t* count_out, void* ws,
                              size_t ws_bytes, void* stream);

/* ---- kernel-level entry points (used by the parity tests; same kernels the calls above launch) ---- */
/* C(M,N) = A(M,K) W(N,K)^T (+bias) (+resid) ; epilogue: 0 bias, 1 bias+resid, 2 bias+SiLU,
 * 3 GEGLU (W/bias must be value/gate interleaved by rap_geglu_interleave; C is (M,N/2)),
 * 4 head-major qkv scatter ([3][H][M][64]), 5 bias + anchor embedding select, 6 bias+ReLU. */
int rap_gemm_f32(int32_t epilogue, const float* A, int32_t lda, const float* W, int32_t ldw, float* C, int32_t ldc,
                 int32_t M, int32_t N, int32_t K, const float* bias, const float* resid, int32_t ldr,
                 const uint8_t* anchor, const float* anchor_emb, int32_t heads, void* stream);
int rap_geglu_interleave(const float* W, const float* b, float* Wp, float* bp, int32_t inner, int32_t K, void* stream);
/* The work list of one attention launch, as rap_sample / rap_dit_forward build it (once per call, on the device): one item
 * {seg_start, seg_len, q0, 0} (4 x int32) per `block_queries` (256) query rows of every segment of cu_seqlens (nseg + 1 entries),
 * LONGEST SEGMENT FIRST when sort_ws (nseg ints of scratch) is given -- a block streams all keys of its segment, blocks are
 * dispatched in order, so the long blocks of a ragged batch (reference: 200 ... 40 000 points per part) must not start last --
 * segment order when sort_ws is NULL; items beyond the last one are zero (seg_len 0 = no work).  items_out: max_items x 4 int32. */
int rap_build_attention_worklist(const int32_t* cu_seqlens, int32_t nseg, int32_t block_queries, int32_t* items_out,
                                 int32_t max_items, int32_t* sort_ws, void* stream);
/* flash_attn_varlen_qkvpacked_func equivalent (reference layer.py:106-111,123-128) on head-major qkv
 * [3][H][TP][64]; out (TP, H*64).  ws >= rap_attention_workspace_bytes(TP, nseg). */
size_t rap_attention_workspace_bytes(int64_t TP, int32_t nseg);
int rap_attention_f32(const float* qkv_headmajor, const int32_t* cu_seqlens, int32_t nseg, float* out, int64_t TP,
                      int32_t heads, const float* logit_bound /* NULL or H device floats, see rap_attention_h16; each must be
                      <= 40 (the fp32 kernel then drops the softmax offset: |score| log2 e <= 58); a head whose bound is larger gets
                      NaN outputs */, void* ws,
                      size_t ws_bytes, void* stream);
int rap_layernorm_mod(const float* x, float* out, int64_t TP, int32_t d, const float* mod, int64_t mod_stride,
                      const int32_t* token_row, void* stream);
int rap_layernorm_affine(const float* x, float* out, int64_t TP, int32_t d, const float* gain, const float* shift,
                         void* stream);
int rap_qknorm(float* qkv_headmajor, int64_t TP, int32_t heads, const float* gamma_q, const float* gamma_k, void* stream);
int rap_posenc_x(const float* x, float* ax, int64_t TP, void* stream);
int rap_posenc_static(const float* cond, const float* scales, const int32_t* token_sample, const float* feat,
                      int32_t feat_dim, float* astatic, int64_t TP, void* stream);
int rap_token_sample(const int32_t* cu_batch, int32_t B, int32_t* token_sample, void* stream);
/* adaLN (scale|shift) table for model `m`: t (rows,), out (rows, 2*num_layers, 2*embed_dim);
 * scratch >= rows*(256 + 4*num_layers*embed_dim) floats. */
int rap_adaln_table(const rap_model* m, const float* t, int32_t rows, float* scratch, float* out, void* stream);


/* ---- reduced-precision kernel-level entry points (dtype 1 = bf16, 2 = fp16; 16-bit tensors as uint16_t*) ---- */
int rap_convert_h16(int32_t dtype, const float* src, uint16_t* dst, int64_t n, void* stream);
/* C = A (M,K) W(N,K)^T, fp32 accumulate.  epilogue: 0 C half = acc + bias; 1 C fp32 = (resid +) acc + bias;
 * 7 C fp16 = fp16(resid + acc + bias) with `resid` pointing at an FP16 (M,N) matrix (required; row stride ldr; may alias C): the residual GEMM
 * of the 16-bit residual stream, one rounding of the fp32 sum (saturating at +-65504), fp16 whatever the operand dtype
 * (6 -- this epilogue's number before ABI version 4, and a different epilogue before that -- is refused);
 * 3 GEGLU on value/gate-interleaved W (C half (M,N/2)); 4 qkv split: q,k -> C half [2][H][M][64], v -> vt, the
 * TRANSPOSED image [H][vt_nblk][64 d][64 pos] the attention kernel consumes: token t sits in block t >> 6 at
 * pos = (t & 51) | ((t & 4) << 1) | ((t & 8) >> 1); vt_nblk * 64 >= M rounded up to 256; rows >= M are written as 0. */
int rap_gemm_h16(int32_t dtype, int32_t epilogue, const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, void* C,
                 int32_t ldc, int32_t M, int32_t N, int32_t K, const float* bias, const float* resid, int32_t ldr,
                 int32_t heads, uint16_t* vt, int32_t vt_nblk, void* stream);
/* Epilogues 1 and 7 of rap_gemm_h16 with the split-K form rap_sample uses for the K = 4d feed-forward GEMM of few-token calls (tuning
 * key 6): when rap_gemm_h16_splitk_workspace_bytes(M, N, K) > 0 (K >= 1024 and at most 128 tiles of 128 x 128), K is split over 2 or 4
 * blocks per tile that write fp32 partial tiles to ws, and a combine pass forms resid + (bias + sum of partials) in a fixed order
 * (deterministic; equals the unsplit result up to the fp32 association of the k-sum).  resid: fp32 (epilogue 1, may be NULL) or fp16
 * (epilogue 7, required) (M,N) matrix with row stride ldr, may alias C.  With a workspace size of 0 the call is rap_gemm_h16. */
size_t rap_gemm_h16_splitk_workspace_bytes(int32_t M, int32_t N, int32_t K);
int rap_gemm_h16_splitk(int32_t dtype, int32_t epilogue, const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, void* C,
                        int32_t ldc, int32_t M, int32_t N, int32_t K, const float* bias, const void* resid, int32_t ldr, void* ws,
                        size_t ws_bytes, void* stream);
/* flash_attn_varlen_qkvpacked_func equivalent on 16-bit operands (fp32 softmax): qk [2][H][TP][64], vt as above,
 * out half (TP, H*64).  ws >= rap_attention_workspace_bytes(TP, nseg).
 * logit_bound: NULL, or H device floats B[h] with q.k/8 <= B[h] <= 40 for every query/key pair of head h GUARANTEED by
 * the caller (after the reference's qk-norm B = 8 max|gamma_q| max|gamma_k|): selects the bounded-softmax kernel
 * (p = exp(s - B), no running maximum; bf16 only -- other dtypes ignore it).  Same softmax, different evaluation order.
 * (Inside rap_sample / rap_dit_forward the bf16 path additionally has qk-norm write q pre-scaled by log2(e)/8 and drops the
 * offset: p = exp2(q'.k); this entry point keeps the un-scaled q convention.) */
/* The QKV projection with the reference's MultiHeadRMSNorm (flow_model/norm.py:28-33) fused into its epilogue -- what rap_sample /
 * rap_dit_forward run in the 16-bit modes (tuning key 7 = 0 restores GEMM + rap_qknorm_h16): q, k rows are normalised from the fp32
 * accumulators, multiplied by gamma and by q_mul (q) or 8 (k), then rounded ONCE to 16 bit into qk_out [2][H][M][64]; v goes to the
 * transposed image vt exactly as epilogue 4 of rap_gemm_h16.  N = 3 * heads * 64, K % 64 == 0, K >= 128. */
int rap_gemm_h16_qkvnorm(int32_t dtype, const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, uint16_t* qk_out, int32_t M,
                         int32_t K, int32_t heads, const float* gamma_q, const float* gamma_k, float q_mul, uint16_t* vt,
                         int32_t vt_nblk, void* stream);
int rap_attention_h16(int32_t dtype, const uint16_t* qk, const uint16_t* vt, int32_t vt_nblk, const int32_t* cu_seqlens,
                      int32_t nseg, uint16_t* out, int64_t TP, int32_t heads, const float* logit_bound, void* ws,
                      size_t ws_bytes, void* stream);
int rap_layernorm_mod_h16(int32_t dtype, const float* x, uint16_t* out, int64_t TP, int32_t d, const float* mod,
                          int64_t mod_stride, const int32_t* token_row, void* stream);
int rap_layernorm_affine_h16(int32_t dtype, const float* x, uint16_t* out, int64_t TP, int32_t d, const float* gain,
                             const float* shift, void* stream);
int rap_qknorm_h16(int32_t dtype, uint16_t* qk, int64_t TP, int32_t heads, const float* gamma_q, const float* gamma_k,
                   void* stream);

/* ---- split-precision kernel-level entry points (compute dtype 3, round 5) ----
 * PAIRED layout of a split fp32 matrix (rows, K): (rows, 2K) fp16; logical column k sits at physical column 64 (k >> 5) + (k & 31) as
 * its fp16 head, and 32 columns further as its fp16 tail -- every 32-column chunk is one 128-byte line [32 heads | 32 tails].  K % 32 == 0.
 * rap_x2_pack:   dst = split(src * scale)  (src fp32 (rows, cols), row stride ld_src; scale a power of two for weights, 1 otherwise).
 * rap_x2_unpack: dst fp32 (rows, cols) = (head + tail) * inv_scale.
 * rap_x2_gemm:   C = A (M,K) W(N,K)^T from paired A (M, lda) and W (N, ldw), K_physical = 2 K (>= 128, % 64 == 0), lda, ldw >= K_physical and % 8 == 0,
 *   N % 256 == 0 (anything else: RAP_ERR_INVALID before any launch); the accumulators are
 *   multiplied by acc_scale (the inverse of the weight planes' scale) before the epilogue:
 *     1  C fp32 (M,N) = resid + acc + bias (resid may be NULL / alias C; ldc, ldr >= N and % 4 == 0: rows move as 16-byte pieces);
 *     3  GEGLU on value/gate-interleaved W: C paired (M, ldc >= N, ldc % 8 == 0) holds the N/2 outputs (h + bh) * gelu_erf(g + bg);
 *     5  QKV projection with MultiHeadRMSNorm fused (N = 3 * heads * 64): q, k -> C paired [2][heads][2 chunks][M][64 physical] (chunk c =
 *        head dims 32c .. 32c+31; q multiplied by q_mul, k by 8 as in rap_gemm_h16_qkvnorm; gamma_q = gamma_k = NULL: no norm, q and k
 *        leave as projected -- qk_norm = False), v -> vt paired
 *        [heads][vt_nblk][2 chunks][64 d][64 physical]: token t sits in block t >> 6, chunk (t >> 5) & 1, at in-chunk position
 *        p = (t & 19) | ((t & 4) << 1) | ((t & 8) >> 1) as head (column p) and tail (column 32 + p); vt_nblk * 64 >= M rounded up to 256.
 * rap_x2_attention: flash_attn_varlen_qkvpacked_func on those planes (online softmax in fp32): out paired (TP, 2 * heads * 64).
 *   ws >= rap_attention_workspace_bytes(TP, nseg).
 * LayerNorm with paired output: rap_layernorm_mod_h16 / rap_layernorm_affine_h16 with dtype 3 (out is (TP, 2 d)). */
int rap_x2_pack(const float* src, int64_t ld_src, int64_t rows, int32_t cols, float scale, uint16_t* dst, void* stream);
int rap_x2_unpack(const uint16_t* src, int64_t rows, int32_t cols, float inv_scale, float* dst, void* stream);
int rap_x2_gemm(int32_t epilogue, const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, void* C, int32_t ldc, int32_t M, int32_t N,
                int32_t K_physical, const float* bias, const float* resid, int32_t ldr, float acc_scale, int32_t heads, const float* gamma_q,
                const float* gamma_k, float q_mul, uint16_t* vt, int32_t vt_nblk, void* stream);
int rap_x2_attention(const uint16_t* qk, const uint16_t* vt, int32_t vt_nblk, const int32_t* cu_seqlens, int32_t nseg, uint16_t* out,
                     int64_t TP, int32_t heads, void* ws, size_t ws_bytes, void* stream);

/* ---- input side of the boundary: raw multi-part scans -> the packed batch rap_sample consumes ----
 * Replaces, in their evaluation-split form (no augmentation), PointCloudDataset._transform
 * (rectified_point_flow/data/dataset.py:733-900) and variable_collate_fn (data/datamodule.py:169-198) for a whole batch, on the
 * device: per sample the frame of the largest ("primary" = anchor) part -- centred on its centroid, scaled by 1.5 x its largest
 * |coordinate| -- then global centring; per part the centred cloud (cond), its pose (R = I, t = centroid; anchor: t = -gt_trans),
 * the anchor masks, and the collated cu_seqlens.  fp64 arithmetic like numpy, fp32 results.
 *   points (TP,3) fp32 or fp64 (points_are_f64), parts of a sample contiguous, samples contiguous; points_per_part (B,P) int64 on
 *   the device, 0 = padding; every sample needs at least one point.
 *   order: NULL, or (TP,) int64 -- for every output point its source index INSIDE its part (what np.random.permutation returned
 *   for that part, dataset.py:819); order_flag: NULL or a device int32 the call ORs 1 into when an index is out of range.
 *   outputs: cond, gt (TP,3) f32; feat_out (TP,F) f32 (feat_in gathered the same way; F may be 0); anchor_indices (TP,) u8;
 *   part_indices (TP,) i64; rotations (B,P,3,3), translations (B,P,3), scales (B,), anchor_parts (B,P) u8,
 *   global_translation (B,3) f32 (mean of the raw sample, original units), cu_seqlens (B+1,) i64.
 *   ws >= rap_collate_workspace_bytes(B, P).  Three launches, no host synchronisation. */
size_t rap_collate_workspace_bytes(int32_t B, int32_t P);
int rap_collate_transform(const void* points, int32_t points_are_f64, const int64_t* points_per_part, int32_t B, int32_t P,
                          int64_t TP, const int64_t* order, const float* feat_in, int32_t F, float* cond, float* gt,
                          float* feat_out, uint8_t* anchor_indices, int64_t* part_indices, float* rotations, float* translations,
                          float* scales, uint8_t* anchor_parts, float* global_translation, int64_t* cu_seqlens,
                          int32_t* order_flag, void* ws, size_t ws_bytes, void* stream);

/* Statistical outlier removal in front of FPS / MiniSpinNet: Open3D's PointCloud.remove_statistical_outlier(nb_neighbors, std_ratio)
 * as dataset_process/extract_sample_features.py:378-385 calls it (20, 2.5).  Open3D is not in the reference mount; the kernel follows
 * its published rule (mean distance to the nb_neighbors nearest points INCLUDING the point itself; threshold = mean + std_ratio *
 * Bessel-corrected std over the cloud; inlier = 0 < d < threshold): parity unpinned.  points (N,3) f32; nb_neighbors <= 32;
 * inlier_indices (>= N int64, ascending), count_out (device int32), stats_out NULL or 3 device doubles {mean, std, threshold};
 * ws >= rap_outlier_workspace_bytes(N).  Brute-force k-NN through the LDS (N^2 pair distances), no host synchronisation. */
size_t rap_outlier_workspace_bytes(int64_t N);
int rap_statistical_outliers(const float* points, int64_t N, int32_t nb_neighbors, double std_ratio, int64_t* inlier_indices,
                             int32_t* count_out, double* stats_out, void* ws, size_t ws_bytes, void* stream);

/* Consistency of a packed batch -- what the reference asserts in split_parts (utils/point_clouds.py:33-52).  rap_sample and the
 * Procrustes entry points REQUIRE sum(points_per_part) == TP and, per sample, sum_p points_per_part[b][p] == cu_seqlens[b+1] -
 * cu_seqlens[b]; they do not read anything back to check it (rap_sample clamps the part table to TP, so a malformed batch gives
 * wrong poses, not an out-of-bounds access).  This entry point writes the verdict to a device int32: 0 = consistent; bit 0 sum
 * != TP, bit 1 cu_seqlens ends, bit 2 cu_seqlens decreasing, bit 3 a sample's parts vs its span, bit 4 a negative size. */
int rap_check_batch(const int64_t* points_per_part, const int32_t* cu_batch, int32_t B, int32_t P, int64_t TP, int32_t* flag_out,
                    void* stream);
/* The deferred form of that check (no host read-back on the call path): enqueue rap_check_batch, the sampling call, then
 * rap_poison_on_flag on every result buffer -- if *flag (device int32, as rap_check_batch wrote it) is non-zero, buf[0..n) is
 * overwritten with NaN, so results of an inconsistent batch cannot be mistaken for poses.  rap_amd.RectifiedPointFlow does this by
 * default and raises on the host the first time it can read the flag without stalling the stream. */
int rap_poison_on_flag(const int32_t* flag, float* buf, int64_t n, void* stream);

/* ---- measurement hooks (bench.py roofline leg) ----
 * When enabled, every attention and layer-GEMM launch inside rap_dit_forward / rap_sample is bracketed by two
 * hipEvents recorded on the launch stream.  rap_profile_collect synchronises on them and returns, per class
 * (0 attention per part, 1 attention per sample, 2 layer GEMMs), the summed milliseconds and the launch count
 * into HOST arrays of 3 entries.  rap_profile_collect_ex (round 5) takes n_classes entries and also reports the HBM-bound ring:
 * 3 LayerNorm, 4 posenc(x_t), 5 Euler update, 6 Procrustes moments + solve, 7 rigid apply / blend (8 classes in all).
 * Off by default; the state is process-global behind a mutex (safe to call from several threads, the figures are per process). */
int rap_profile_enable(int on);
int rap_profile_collect_ex(float* h_ms_out, int64_t* h_count_out, int32_t n_classes);
/* Production switches between two SHIPPED code paths that compute the same function (process-global, atomic; not per model):
 *   key 5  split-KV attention for few-token calls        {0 off, 1 on (default)}       fp32 path and split precision
 *   key 6  split-K of the bias + residual GEMM, few rows  {0 off, 1 on (default)}       every precision (16-bit / split precision: K >= 1024, i.e. ff2)
 *   key 7  qk-norm fused into the QKV GEMM epilogue       {1 (default), 0 = own kernel} both precisions
 *   key 9  GEGLU's Phi                                    {1 (default): erfc polynomial, |error| <= 1.5e-7; 0: erff}   fp32 path
 *   key 11 persistent 16-bit GEMM (one block per CU walks {1 (default), 0 = one 256 x 256 tile per block}   16-bit path
 *          its XCD's tiles; full-tile shapes only)
 *   key 12 persistent fp32 GEMM (same, 256 x 256 kernel)  {1 (default), 0 = one tile per block}              fp32 path
 *   key 13 16-bit attention: K / V^T tiles by LDS-DMA      {1 (default), 0 = staged through registers}       16-bit path
 *   key 15 attention work lists of rap_sample / forward    {1 (default): longest segment first, 0: segment order}   both precisions
 *   key 16 split-precision attention: blocks per CU          {2 (default): one 8-wave block, 4: two}                 compute dtype 3
 *   key 17 split precision from this many token rows per call {1024 (default); 0 = always}: SMALLER calls of a model in compute dtype 3
 *          run the exact-fp32 kernels (both are fp32-accurate; below a few thousand tokens the fp32 path's few-token forms are faster).
 *          rap_workspace_bytes of such a model covers both layouts, so the key may change between the size query and the call
 *   key 18 four-stage LDS-DMA ring of the 128 x 128 16-bit / split-precision GEMM (few-token calls) for launches of at most this many
 *          blocks {256 (default); 0 = never: two stages, two blocks per CU}.  Bit-identical results.
 *   key 19 few-token 16-bit / split-precision calls: the combine pass of every residual GEMM folded into the LayerNorm that follows it
 *          {1 (default), 0 = the round-5 launch sequence}.  Bit-identical in the 16-bit modes; in split precision the fused sequence
 *          also splits K of the out-projection (fp32-class agreement)
 *   key 20 16-bit attention of few-token calls (<= 4 096 token rows)
 *          {1 (default): the eight waves of a block are query waves x KEY GROUPS -- work items of 64 rows x 4 key groups up to 2 048 token
 *             rows, 128 rows x 2 key groups up to 4 096; the groups' partial (O, l, m) meet in LDS, nothing extra leaves the CU.  The keys of a
 *             row are summed in another order than by the unsplit kernel: deterministic, same error bound, not bit-identical to it;
 *           2: work items of 128 rows and a four-stage K / V^T ring, no key groups -- bit-identical to 0;
 *           0: 256-row items, two stages (round 5);
 *           64 / 128: that item size + ring, 66 / 130: 64 rows x 4 / 128 rows x 2 key groups -- forced for every call of at most 8 192 token
 *             rows AND for rap_attention_h16 (the A/B values)}.  (rap_workspace_bytes does not depend on it.)
 * Operand range of compute dtype 3 (and of the fp16 residual stream): every paired activation -- LayerNorm output, q / k (also without
 * qk-norm), v, attention output, GEGLU output -- is clipped to +-65 504 before it is split into head and tail (NaN stays NaN), and
 * rap_model_set_compute_dtype(3) refuses weights that are not finite.  scripts/check_checkpoint.py compares the mode with exact fp32 on
 * the weights it is given and exits non-zero on a miss.
 * Any other key returns RAP_ERR_INVALID.  (Keys 0-4 selected among the kernel variants of the round-1/2 experiments; those variants
 * are no longer in the tree, and what is left of the switch exists only in a library built with -DRAP_ABLATION_BUILD.) */
int rap_set_tuning(int32_t key, int32_t value);
int rap_profile_reset(void);
int rap_profile_collect(float* h_ms_out, int64_t* h_count_out);

#ifdef __cplusplus
}
#endif
#endif /* RAPFLOW_H */
