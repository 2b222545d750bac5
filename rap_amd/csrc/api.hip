// C ABI of librapflow (see include/rapflow.h): model packing, workspace carving and the launch
// sequences of one velocity-network forward and of the whole Euler sampling loop.  Host code only;
// every kernel lives in the sibling .hip files.
#include "../../include/rapflow.h"
#include "half.h"
#include "kernels.h"

#include <mutex>
#include <new>
#include <vector>

static thread_local int g_last_hip_error = 0;
void rap_set_last_hip_error(int e) { g_last_hip_error = e; }

// ---------------------------------------------------------------------------------------------
// optional per-kernel-class timing with HIP events on the launch stream (bench.py's roofline leg).
// Off by default; when on, every attention / GEMM launch of forward_step is bracketed by two events.
// ---------------------------------------------------------------------------------------------
// classes: 0 = attention per part, 1 = attention per sample, 2 = layer GEMMs (the MFMA-bound kernels), and since round 5 the HBM-bound
// ring: 3 = LayerNorm, 4 = posenc(x_t), 5 = Euler update, 6 = Procrustes moments + solve, 7 = rigid apply / blend
#define RAP_PROF_CLASSES 8
struct ProfRec { hipEvent_t a, b; int cls; };
// process-global state behind one mutex (round 5: the enable flag was a plain bool and the vectors unsynchronised): two
// threads enqueueing on two models may record concurrently; the flag is an atomic so that the off path costs one relaxed load
static std::atomic<bool> g_prof_on{false};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;
static size_t g_prof_pool_used = 0;
static const size_t RAP_PROF_MAX_EVENTS = 1u << 19;   // 20 sample calls of rap_12 at 20 flow steps = 38 400 scopes = 76 800 events (r01: 32768 truncated the log)

static hipEvent_t prof_event() {      // caller holds g_prof_mu
  if (g_prof_pool_used < g_prof_pool.size()) return g_prof_pool[g_prof_pool_used++];
  if (g_prof_pool.size() >= RAP_PROF_MAX_EVENTS) return nullptr;
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  g_prof_pool.push_back(e);
  g_prof_pool_used++;
  return e;
}
struct ProfScope {
  hipStream_t s; hipEvent_t a = nullptr, b = nullptr; int cls;
  ProfScope(hipStream_t s_, int cls_) : s(s_), cls(cls_) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    a = prof_event(); b = prof_event();
    if (a && b) (void)hipEventRecord(a, s); else a = b = nullptr;
  }
  ~ProfScope() {
    if (a && b) {
      std::lock_guard<std::mutex> lock(g_prof_mu);
      (void)hipEventRecord(b, s); g_prof_recs.push_back({a, b, cls});
    }
  }
};
extern "C" int rap_profile_enable(int on) { g_prof_on.store(on != 0); return RAP_OK; }
extern "C" int rap_profile_reset(void) { std::lock_guard<std::mutex> lock(g_prof_mu); g_prof_recs.clear(); g_prof_pool_used = 0; return RAP_OK; }
// Synchronises on the recorded events.  h_ms_out / h_count_out have n_classes entries (classes >= n_classes are dropped).
extern "C" int rap_profile_collect_ex(float* h_ms_out, int64_t* h_count_out, int32_t n_classes) {
  if (!h_ms_out || !h_count_out || n_classes <= 0) return RAP_ERR_INVALID;
  std::lock_guard<std::mutex> lock(g_prof_mu);
  for (int c = 0; c < n_classes; ++c) { h_ms_out[c] = 0.f; h_count_out[c] = 0; }
  for (const ProfRec& r : g_prof_recs) {
    if (r.cls >= n_classes) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) return RAP_ERR_HIP;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return RAP_ERR_HIP;
    h_ms_out[r.cls] += ms; h_count_out[r.cls] += 1;
  }
  return RAP_OK;
}
// the three MFMA-bound classes (ABI <= 4 form)
extern "C" int rap_profile_collect(float* h_ms_out, int64_t* h_count_out) { return rap_profile_collect_ex(h_ms_out, h_count_out, 3); }

// ---------------------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------------------
struct LayerW {
  const float* Wqkv[2];   // (3d,d)  [0]=self (per part), [1]=global (per sample)
  const float* Wout[2];   // (d,d)
  const float* bout[2];   // (d)
  const float* gq[2];     // (H,64)
  const float* gk[2];
  const float* ffn_g;     // (d)
  const float* ffn_b;
  const float* Wff1p;     // (8d,d) value/gate interleaved
  const float* bff1p;     // (8d)
  const float* Wff2;      // (d,4d)
  const float* bff2;      // (d)
};

// 16-bit copies of the transformer-block GEMM weights (one set per reduced-precision dtype, built on demand)
struct LayerWH {
  const u16* Wqkv[2];
  const u16* Wout[2];
  const u16* Wff1p;   // value/gate interleaved, as the fp32 packing
  const u16* Wff2;
  // split precision (RAP_DT_F32X2): the planes hold w * 2^e per tensor; s_* = 2^-e, the factor the epilogue applies to the accumulators
  float s_qkv[2] = {1.f, 1.f}, s_out[2] = {1.f, 1.f}, s_ff1 = 1.f, s_ff2 = 1.f;
  // ... and the powers of two the ACTIVATIONS those weights produce are stored with (round 6, GemmParamsH::out_scale): g_v for the V^T image
  // (2^(e_qkv - 16): 1 for a tensor whose largest weight lies in (2^-5, 2^-4], the nn.Linear default range at d = 512), g_ff1 for the GEGLU
  // output (value x gate: 2^(2 (e_ff1 - 16))).  The consuming GEMM (out-projection, ff2) divides its accumulator scale by it (exact).
  float g_v[2] = {1.f, 1.f}, g_ff1 = 1.f;
};
struct HalfWeights {
  u16* blob = nullptr;
  std::vector<LayerWH> layers;
};
#define RAP_MAX_LOGIT_BOUND 40.0f   // exp(-2*40) is still a normal fp32 / bf16 number

struct rap_model {
  rap_model_desc desc;
  int d, L, H, F, E;
  int Din = 0;                // in_dim: width of the latent point features concatenated into the embedding (embedding.py:163-166; 0 in every shipped config)
  int Ks = 128;               // columns of the step-invariant embedding input: 128 + align_up(Din, 32)
  int dtype = RAP_DT_F32;     // arithmetic type of the transformer blocks (rap_model_set_compute_dtype)
  bool qk_norm = true;        // MultiHeadRMSNorm on q / k (rap_model_set_qk_norm; every shipped configuration has it on)
  int resid_dtype = RAP_DT_F32;   // storage type of the residual stream in the 16-bit modes (rap_model_set_residual_dtype): fp32 or fp16
  HalfWeights half[4];        // indexed by dtype (slot 0 unused; 3 = the paired head / tail planes of the split-precision mode)
  float* logit_bound = nullptr;   // (L, 2, H) per-head bounds on q.k/8 after qk-norm; null until a 16-bit dtype is selected
  // bounded[2 * layer + branch]: every head of THAT attention has a bound <= RAP_MAX_LOGIT_BOUND, so that launch may use the
  // bounded (offset-free) softmax kernel; the others take the online-softmax kernel.  Decided per launch since round 3 (VERDICT
  // r02 item 4): one hot head in a trained checkpoint costs its own (layer, branch) the fast kernel, not the whole model.
  std::vector<uint8_t> bounded;
  int n_bounded = 0;
  float* raw = nullptr;       // copy of the caller's blob
  float* derived = nullptr;   // packed arrays
  const float* anchor_emb;    // (2,d)
  const float* emb_bias;      // (d)
  const float* Wstatic;       // (d,Ks): [cond PE63 | scale PE21 | feat F | 0 .. 128 | latent Din | 0]
  const float* Wx;            // (d,64):  [x_t PE63 | 0]
  const float *adaW1, *adab1, *adaW2, *adab2, *adaW3, *adab3;  // stacked over j = 2*layer + {0 self, 1 global}
  std::vector<LayerW> layers;
  const float *hW0, *hb0, *hW2, *hb2, *hW4;
  // Configuration (compute / residual dtype) is read on the HOST while a call is being enqueued (workspace layout, kernel choice):
  // the setters and the enqueueing entry points take this mutex, so a setter on one thread cannot change the layout of a call another
  // thread is in the middle of enqueueing.  Work that is already enqueued is not affected by a later change.
  mutable std::mutex cfg_mu;
};

static bool desc_ok(const rap_model_desc* d) {
  if (!d) return false;
  if (d->embed_dim <= 0 || d->embed_dim % 256 != 0 || d->embed_dim > 1024) return false;
  if (d->num_heads * 64 != d->embed_dim) return false;
  if (d->num_layers <= 0 || d->num_layers > 64) return false;
  if (d->local_feat_dim < 0 || d->local_feat_dim % 4 != 0 || d->local_feat_dim > 40) return false;
  return true;
}

extern rap_tuning_t g_rap_gemm_variant;   // gemm_f32.hip
extern rap_tuning_t g_rap_gemm_stagger;   // gemm_f32.hip
extern rap_tuning_t g_rap_gemm_splitk;    // gemm_f32.hip
extern rap_tuning_t g_rap_geglu_fast;     // gemm_f32.hip
extern rap_tuning_t g_rap_attn_split;     // attn_f32.hip
extern rap_tuning_t g_rap_gemm_h16_variant;   // gemm_h16.hip
extern rap_tuning_t g_rap_attn_h16_variant;   // attn_h16.hip
extern rap_tuning_t g_rap_gemm_h16_persistent;   // gemm_h16.hip
extern rap_tuning_t g_rap_attn_h16_dma;          // attn_h16.hip
extern rap_tuning_t g_rap_gemm_f32_persistent;   // gemm_f32.hip
extern rap_tuning_t g_rap_attn_x2_wpe;           // attn_x2.hip
extern rap_tuning_t g_rap_attn_h16_small;        // attn_h16.hip, tuning key 20
rap_tuning_t g_rap_attn_lpt = 1;               // tuning key 15: attention work lists longest-segment-first (1, default) or in segment order (0)
// tuning key 17: split precision takes over from this many token rows (align_up(TP, 256)) per call; SMALLER calls of a model in compute
// dtype 3 run the exact-fp32 kernels -- both are fp32-accurate, and below a few thousand tokens every kernel of a layer sits at the launch
// floor, where the fp32 path's few-token forms (128 x 128 tiles, split-K, split-KV) are the faster ones (r05 call 3: configs[0] geometry,
// one pair of 2 x 1024 points: 48.3 ms fp32 vs 52.6 ms on the 256 x 256-tile split kernels; call 4, one sample of 8 views, 20 steps:
// 1 024 tokens 71 vs 89 ms, 2 048: 92 vs 99, 4 096: 197 vs 124, 8 000: 459 vs 197; call 5, with the residual GEMMs of few-token calls on
// 128 x 128 split-precision tiles: 2 048 tokens 92 vs 81 ms, 2 560: 149 vs 86, 3 072: 169 vs 91; call 9, with every few-token GEMM on
// 128 x 128 tiles, split-K of ff2 and split-KV attention: 1 024 tokens 71 vs 46 ms, 2 048: 92 vs 54, configs[0] geometry 48.4 vs 28.0 ms --
// split precision wins at every size measured; calls below 1 024 rows (not measured) stay on the fp32 kernels).  rap_workspace_bytes of a
// split-precision model covers both layouts, so the key may change between the size query and the call.
rap_tuning_t g_rap_x2_min_rows = 1024;
rap_tuning_t g_rap_small_fused = 1;            // tuning key 19: few-token 16-bit / split-precision calls fold every residual GEMM's combine pass into the following LayerNorm (1, default)
extern rap_tuning_t g_rap_ring_blocks;         // gemm_h16.hip, tuning key 18: launches of the 128 x 128 16-bit GEMM with at most this many blocks take the four-stage ring
rap_tuning_t g_rap_fuse_qknorm = 1;            // tuning key 7: qk-norm fused into the QKV GEMM epilogue (1, default; both precisions) or as its own kernel (0)
// Production switches (process-global, atomics): each selects between two SHIPPED code paths that produce the same result up to
// fp32 summation order -- 5 split-KV for few-token calls (fp32 attention), 6 split-K for few-row calls (fp32 GEMMs and the 16-bit
// residual GEMMs), 7 fused qk-norm, 9 GEGLU's Phi by the
// 1.5e-7 erfc polynomial (1) or erff (0), 11 / 12 persistent 16-bit / fp32 GEMM, 15 the order of the attention work lists (identical results).  Keys 0-4 (kernel-variant A/B of the round-1/2 experiments) exist
// only in a library built with -DRAP_ABLATION_BUILD; the shipped library refuses them.
extern "C" int rap_set_tuning(int32_t key, int32_t value) {
#ifdef RAP_ABLATION_BUILD
  if (key == 0 && (value == 16 || value == 32 || value == 48)) { g_rap_gemm_variant = value; return RAP_OK; }
  if (key == 2 && (value == 0 || value == 1 || value == 14)) { g_rap_gemm_h16_variant = value; return RAP_OK; }
  if (key == 3 && (value == 0 || value == 5)) { g_rap_attn_h16_variant = value; return RAP_OK; }
  if (key == 4 && value >= 0 && value <= 2) { g_rap_gemm_stagger = value; return RAP_OK; }
#endif
  if (key == 5 && (value == 0 || value == 1)) { g_rap_attn_split = value; return RAP_OK; }
  if (key == 6 && (value == 0 || value == 1)) { g_rap_gemm_splitk = value; return RAP_OK; }
  if (key == 7 && (value == 0 || value == 1)) { g_rap_fuse_qknorm = value; return RAP_OK; }
  if (key == 9 && (value == 0 || value == 1)) { g_rap_geglu_fast = value; return RAP_OK; }
  if (key == 11 && (value == 0 || value == 1)) { g_rap_gemm_h16_persistent = value; return RAP_OK; }
  if (key == 12 && (value == 0 || value == 1)) { g_rap_gemm_f32_persistent = value; return RAP_OK; }
  if (key == 13 && (value == 0 || value == 1)) { g_rap_attn_h16_dma = value; return RAP_OK; }
  if (key == 15 && (value == 0 || value == 1)) { g_rap_attn_lpt = value; return RAP_OK; }
  if (key == 17 && value >= 0) { g_rap_x2_min_rows = value; return RAP_OK; }   // split precision from this many token rows per call (smaller calls: exact fp32)
  if (key == 18 && value >= 0) { g_rap_ring_blocks = value; return RAP_OK; }      // four-stage ring of the 128 x 128 16-bit GEMM up to this many blocks per launch (0 = never)
  if (key == 19 && (value == 0 || value == 1)) { g_rap_small_fused = value; return RAP_OK; }      // combine + LayerNorm fusion of few-token calls
  if (key == 20 && (value == 0 || value == 1 || value == 2 || value == 64 || value == 128 || value == 66 || value == 130)) { g_rap_attn_h16_small = value; return RAP_OK; }   // 16-bit attention of few-token calls: 64 / 128-row work items + four-stage ring
  if (key == 16 && (value == 2 || value == 4)) { g_rap_attn_x2_wpe = value; return RAP_OK; }   // split-precision attention: 1 / 2 blocks per CU
  return RAP_ERR_INVALID;
}

extern "C" int rap_version(void) { return RAPFLOW_ABI_VERSION; }
extern "C" int rap_last_hip_error(void) { return g_last_hip_error; }

static bool in_dim_ok(int32_t in_dim) { return in_dim >= 0 && in_dim <= 512 && in_dim % 4 == 0; }
extern "C" int64_t rap_weight_count(const rap_model_desc* desc) { return rap_weight_count_latent(desc, 0); }
extern "C" int64_t rap_weight_count_latent(const rap_model_desc* desc, int32_t in_dim) {
  if (!desc_ok(desc) || !in_dim_ok(in_dim)) return -1;
  const int64_t d = desc->embed_dim, L = desc->num_layers, H = desc->num_heads, E = 147 + desc->local_feat_dim + in_dim;
  int64_t n = 2 * d + d * E + d;
  const int64_t attn = (d * 256 + d) + (d * d + d) + (2 * d * d + 2 * d) + 3 * d * d + (d * d + d) + 2 * H * 64;
  const int64_t layer = 2 * attn + 2 * d + (8 * d * d + 8 * d) + (4 * d * d + d);
  n += L * layer;
  n += (d * d + d) + (d / 2 * d + d / 2) + 3 * (d / 2);
  return n;
}

static int ensure_logit_bounds(rap_model* m, hipStream_t stream);

extern "C" int rap_model_create(const rap_model_desc* desc, const float* d_weights, int64_t n_floats, void* stream_,
                                rap_model** out) {
  return rap_model_create_latent(desc, 0, d_weights, n_floats, stream_, out);
}
extern "C" int rap_model_create_latent(const rap_model_desc* desc, int32_t in_dim, const float* d_weights, int64_t n_floats, void* stream_,
                                       rap_model** out) {
  if (!out) return RAP_ERR_INVALID;
  *out = nullptr;
  if (!desc_ok(desc) || !in_dim_ok(in_dim) || !d_weights) return RAP_ERR_INVALID;
  if (n_floats != rap_weight_count_latent(desc, in_dim)) return RAP_ERR_INVALID;
  hipStream_t stream = (hipStream_t)stream_;
  rap_model* m = new (std::nothrow) rap_model();
  if (!m) return RAP_ERR_ALLOC;
  m->desc = *desc;
  const int d = m->d = desc->embed_dim, L = m->L = desc->num_layers, H = m->H = desc->num_heads;
  m->F = desc->local_feat_dim;
  m->Din = in_dim;
  const int Ks = m->Ks = 128 + (in_dim + 31) / 32 * 32;
  const int E = m->E = 147 + m->F + m->Din;      // native column order: [cond 63 | x_t 63 | scale 21 | feat F | latent Din]
  if (hipMalloc((void**)&m->raw, (size_t)n_floats * sizeof(float)) != hipSuccess) { delete m; return RAP_ERR_ALLOC; }
  const size_t n_ada = (size_t)2 * L * ((size_t)d * 256 + d + (size_t)d * d + d + (size_t)2 * d * d + 2 * d);
  const size_t n_ff1 = (size_t)L * ((size_t)8 * d * d + 8 * d);
  const size_t n_derived = (size_t)d * Ks + (size_t)d * 64 + n_ada + n_ff1;
  if (hipMalloc((void**)&m->derived, n_derived * sizeof(float)) != hipSuccess) {
    (void)hipFree(m->raw); delete m; return RAP_ERR_ALLOC;
  }
  int rc = RAP_OK;
  auto fail = [&](int code) { rap_model_destroy(m); return code; };
  if (hipMemcpyAsync(m->raw, d_weights, (size_t)n_floats * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess)
    return fail(RAP_ERR_HIP);

  // ---- walk the raw blob in state_dict order
  const float* p = m->raw;
  auto take = [&](size_t n) { const float* r = p; p += n; return r; };
  m->anchor_emb = take((size_t)2 * d);
  const float* embW = take((size_t)d * E);
  m->emb_bias = take(d);

  float* q = m->derived;
  auto carve = [&](size_t n) { float* r = q; q += n; return r; };
  float* Wstatic = carve((size_t)d * Ks);
  float* Wx = carve((size_t)d * 64);
  float* aW1 = carve((size_t)2 * L * d * 256); float* ab1 = carve((size_t)2 * L * d);
  float* aW2 = carve((size_t)2 * L * d * d);   float* ab2 = carve((size_t)2 * L * d);
  float* aW3 = carve((size_t)2 * L * 2 * d * d); float* ab3 = carve((size_t)2 * L * 2 * d);
  m->Wstatic = Wstatic; m->Wx = Wx;
  m->adaW1 = aW1; m->adab1 = ab1; m->adaW2 = aW2; m->adab2 = ab2; m->adaW3 = aW3; m->adab3 = ab3;

  // embedding projection columns (embedding.py:161-177): [0,63) cond | [63,126) x_t | [126,147) scale | [147,E) feat
  if ((rc = launch_fill_zero(stream, Wstatic, (size_t)d * Ks))) return fail(rc);
  if ((rc = launch_fill_zero(stream, Wx, (size_t)d * 64))) return fail(rc);
  if ((rc = launch_copy_cols(stream, embW, E, 0, Wstatic, Ks, 0, d, 63))) return fail(rc);
  if ((rc = launch_copy_cols(stream, embW, E, 126, Wstatic, Ks, 63, d, 21))) return fail(rc);
  if ((rc = launch_copy_cols(stream, embW, E, 147, Wstatic, Ks, 84, d, m->F))) return fail(rc);
  if ((rc = launch_copy_cols(stream, embW, E, 147 + m->F, Wstatic, Ks, 128, d, m->Din))) return fail(rc);      // latent columns: one more k-range of the hoisted GEMM
  if ((rc = launch_copy_cols(stream, embW, E, 63, Wx, 64, 0, d, 63))) return fail(rc);

  m->layers.resize(L);
  for (int i = 0; i < L; ++i) {
    LayerW& lw = m->layers[i];
    for (int a = 0; a < 2; ++a) {
      const int j = 2 * i + a;
      const float* W1 = take((size_t)d * 256); const float* b1 = take(d);
      const float* W2 = take((size_t)d * d);   const float* b2 = take(d);
      const float* W3 = take((size_t)2 * d * d); const float* b3 = take((size_t)2 * d);
      if ((rc = launch_copy_cols(stream, W1, 256, 0, aW1 + (size_t)j * d * 256, 256, 0, d, 256))) return fail(rc);
      if ((rc = launch_copy_cols(stream, b1, d, 0, ab1 + (size_t)j * d, d, 0, 1, d))) return fail(rc);
      if ((rc = launch_copy_cols(stream, W2, d, 0, aW2 + (size_t)j * d * d, d, 0, d, d))) return fail(rc);
      if ((rc = launch_copy_cols(stream, b2, d, 0, ab2 + (size_t)j * d, d, 0, 1, d))) return fail(rc);
      if ((rc = launch_copy_cols(stream, W3, d, 0, aW3 + (size_t)j * 2 * d * d, d, 0, 2 * d, d))) return fail(rc);
      if ((rc = launch_copy_cols(stream, b3, 2 * d, 0, ab3 + (size_t)j * 2 * d, 2 * d, 0, 1, 2 * d))) return fail(rc);
      lw.Wqkv[a] = take((size_t)3 * d * d);
      lw.Wout[a] = take((size_t)d * d);
      lw.bout[a] = take(d);
      lw.gq[a] = take((size_t)H * 64);
      lw.gk[a] = take((size_t)H * 64);
    }
    lw.ffn_g = take(d);
    lw.ffn_b = take(d);
    const float* Wff1 = take((size_t)8 * d * d);
    const float* bff1 = take((size_t)8 * d);
    float* Wp = carve((size_t)8 * d * d);
    float* bp = carve((size_t)8 * d);
    if ((rc = launch_geglu_interleave(stream, Wff1, bff1, Wp, bp, 4 * d, d))) return fail(rc);
    lw.Wff1p = Wp; lw.bff1p = bp;
    lw.Wff2 = take((size_t)d * 4 * d);
    lw.bff2 = take(d);
  }
  m->hW0 = take((size_t)d * d); m->hb0 = take(d);
  m->hW2 = take((size_t)(d / 2) * d); m->hb2 = take(d / 2);
  m->hW4 = take((size_t)3 * (d / 2));
  if ((int64_t)(p - m->raw) != n_floats || (size_t)(q - m->derived) != n_derived) return fail(RAP_ERR_INVALID);
  if ((rc = ensure_logit_bounds(m, stream))) return fail(rc);
  *out = m;
  return RAP_OK;
}

extern "C" void rap_model_destroy(rap_model* m) {
  if (!m) return;
  if (m->raw) (void)hipFree(m->raw);
  if (m->derived) (void)hipFree(m->derived);
  for (int i = 0; i < 4; ++i)
    if (m->half[i].blob) (void)hipFree(m->half[i].blob);
  if (m->logit_bound) (void)hipFree(m->logit_bound);
  delete m;
}

// Per-head bounds on the attention logits from the qk-norm gains (a property of the weights): (L, 2, H) floats on the
// device; read back once (model creation may synchronise) to decide whether the bounded-softmax kernels are admissible.
static int ensure_logit_bounds(rap_model* m, hipStream_t stream) {
  if (m->logit_bound) return RAP_OK;
  const int n = m->L * 2 * m->H;
  if (hipMalloc((void**)&m->logit_bound, (size_t)n * sizeof(float)) != hipSuccess) { m->logit_bound = nullptr; return RAP_ERR_ALLOC; }
  for (int i = 0; i < m->L; ++i)
    for (int a = 0; a < 2; ++a) {
      const int rc = launch_qk_logit_bound(stream, m->layers[i].gq[a], m->layers[i].gk[a], m->H, m->logit_bound + (size_t)(2 * i + a) * m->H);
      if (rc) return rc;
    }
  std::vector<float> hb(n);
  RAP_HIP_CHECK(hipMemcpyAsync(hb.data(), m->logit_bound, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, stream));
  RAP_HIP_CHECK(hipStreamSynchronize(stream));
  m->bounded.assign((size_t)2 * m->L, 1);
  for (int j = 0; j < 2 * m->L; ++j)
    for (int h = 0; h < m->H; ++h)
      if (!(hb[(size_t)j * m->H + h] <= RAP_MAX_LOGIT_BOUND)) m->bounded[j] = 0;
  m->n_bounded = 0;
  for (uint8_t b : m->bounded) m->n_bounded += b;
  return RAP_OK;
}

// Split precision: paired fp16 head / tail planes of the six GEMM weights of every layer, each tensor multiplied by a power of two
// chosen from its largest magnitude (x2_pack.hip).  Model configuration may synchronise (as model creation does for the logit bounds).
static int build_x2_weights(rap_model* m, hipStream_t stream) {
  HalfWeights& hw = m->half[RAP_DT_F32X2];
  const size_t d = m->d, L = m->L;
  const size_t per_layer = 2 * (2 * (3 * d * d + d * d) + 8 * d * d + 4 * d * d);      // twice the 16-bit copy: heads and tails
  const int nt = (int)(6 * L);
  float* d_max = nullptr;
  if (hipMalloc((void**)&d_max, (size_t)nt * sizeof(float)) != hipSuccess) return RAP_ERR_ALLOC;
  if (hipMalloc((void**)&hw.blob, per_layer * L * sizeof(u16)) != hipSuccess) { hw.blob = nullptr; (void)hipFree(d_max); return RAP_ERR_ALLOC; }
  int rc = RAP_OK;
  auto fail = [&](int code) { (void)hipFree(hw.blob); hw.blob = nullptr; hw.layers.clear(); (void)hipFree(d_max); return code; };
  if (hipMemsetAsync(d_max, 0, (size_t)nt * sizeof(float), stream) != hipSuccess) return fail(RAP_ERR_HIP);
  struct Item { const float* src; size_t rows, cols; };
  std::vector<Item> items;
  for (size_t i = 0; i < L; ++i) {
    const LayerW& lw = m->layers[i];
    for (int a = 0; a < 2; ++a) { items.push_back({lw.Wqkv[a], 3 * d, d}); items.push_back({lw.Wout[a], d, d}); }
    items.push_back({lw.Wff1p, 8 * d, d});
    items.push_back({lw.Wff2, d, 4 * d});
  }
  for (int k = 0; k < nt && rc == RAP_OK; ++k) rc = launch_max_abs(stream, items[k].src, items[k].rows * items[k].cols, d_max + k);
  if (rc) return fail(rc);
  std::vector<float> hmax(nt);
  if (hipMemcpyAsync(hmax.data(), d_max, (size_t)nt * sizeof(float), hipMemcpyDeviceToHost, stream) != hipSuccess) return fail(RAP_ERR_HIP);
  if (hipStreamSynchronize(stream) != hipSuccess) return fail(RAP_ERR_HIP);
  u16* q = hw.blob;
  std::vector<const u16*> planes(nt);
  std::vector<float> inv(nt);
  std::vector<int> expo(nt);
  for (int k = 0; k < nt && rc == RAP_OK; ++k) {
    // largest magnitude * 2^e in (2^11, 2^12]; an all-zero tensor keeps e = 0
    // a NaN / Inf weight would silently become +-65 504 in the planes (saturating split): refuse the model instead
    if (!(hmax[k] < 3.0e38f)) return fail(RAP_ERR_INVALID);
    int e = 0;
    if (hmax[k] > 0.f) { int ex; (void)frexpf(hmax[k], &ex); e = 12 - ex; }
    e = e > 60 ? 60 : e < -60 ? -60 : e;
    expo[k] = hmax[k] > 0.f ? e : 16;
    inv[k] = ldexpf(1.0f, -e);
    planes[k] = q;
    rc = launch_x2_pack(stream, items[k].src, (long)items[k].cols, (long)items[k].rows, (int)items[k].cols, ldexpf(1.0f, e), q);
    q += 2 * items[k].rows * items[k].cols;
  }
  if (rc) return fail(rc);
  hw.layers.resize(L);
  for (size_t i = 0; i < L; ++i) {
    LayerWH& lh = hw.layers[i];
    const int k0 = (int)(6 * i);
    for (int a = 0; a < 2; ++a) {
      lh.Wqkv[a] = planes[k0 + 2 * a]; lh.s_qkv[a] = inv[k0 + 2 * a];
      { int g = expo[k0 + 2 * a] - 16; g = g > 12 ? 12 : g < -12 ? -12 : g; lh.g_v[a] = ldexpf(1.0f, g); }
      lh.Wout[a] = planes[k0 + 2 * a + 1]; lh.s_out[a] = inv[k0 + 2 * a + 1];
    }
    lh.Wff1p = planes[k0 + 4]; lh.s_ff1 = inv[k0 + 4];
    { int g = 2 * (expo[k0 + 4] - 16); g = g > 24 ? 24 : g < -24 ? -24 : g; lh.g_ff1 = ldexpf(1.0f, g); }
    lh.Wff2 = planes[k0 + 5]; lh.s_ff2 = inv[k0 + 5];
  }
  (void)hipFree(d_max);
  return RAP_OK;
}

extern "C" int rap_model_set_compute_dtype(rap_model* m, int32_t dtype, void* stream_) {
  if (!m) return RAP_ERR_INVALID;
  std::lock_guard<std::mutex> lock(m->cfg_mu);
  if (dtype == RAP_DT_F32) { m->dtype = dtype; return RAP_OK; }
  if (dtype != RAP_DT_BF16 && dtype != RAP_DT_F16 && dtype != RAP_DT_F32X2) return RAP_ERR_INVALID;
  HalfWeights& hw = m->half[dtype];
  if (!hw.blob && dtype == RAP_DT_F32X2) {
    const int rcx = build_x2_weights(m, (hipStream_t)stream_);
    if (rcx) return rcx;
  }
  if (!hw.blob) {
    hipStream_t stream = (hipStream_t)stream_;
    const size_t d = m->d, L = m->L;
    const size_t per_layer = 2 * (3 * d * d + d * d) + 8 * d * d + 4 * d * d;
    if (hipMalloc((void**)&hw.blob, per_layer * L * sizeof(u16)) != hipSuccess) { hw.blob = nullptr; return RAP_ERR_ALLOC; }
    u16* q = hw.blob;
    int rc = RAP_OK;
    auto conv = [&](const float* src, size_t n) -> const u16* {
      u16* dst = q; q += n;
      if (rc == RAP_OK) rc = launch_convert_h16(stream, dtype, src, dst, n);
      return dst;
    };
    hw.layers.resize(L);
    for (size_t i = 0; i < L; ++i) {
      const LayerW& lw = m->layers[i];
      LayerWH& lh = hw.layers[i];
      for (int a = 0; a < 2; ++a) {
        lh.Wqkv[a] = conv(lw.Wqkv[a], 3 * d * d);
        lh.Wout[a] = conv(lw.Wout[a], d * d);
      }
      lh.Wff1p = conv(lw.Wff1p, 8 * d * d);
      lh.Wff2 = conv(lw.Wff2, 4 * d * d);
    }
    if (rc != RAP_OK) { (void)hipFree(hw.blob); hw.blob = nullptr; hw.layers.clear(); return rc; }
  }
  { const int rcb = ensure_logit_bounds(m, (hipStream_t)stream_); if (rcb) return rcb; }
  m->dtype = dtype;
  return RAP_OK;
}
extern "C" int rap_model_compute_dtype(const rap_model* m) { return m ? m->dtype : RAP_ERR_INVALID; }
// Residual stream of the 16-bit modes: fp32 (default) or fp16 -- what the reference's autocast inference holds (nn.Linear outputs
// are 16-bit under Lightning "16-mixed", layer.py:155-164 adds them).  Ignored while the compute dtype is fp32.
extern "C" int rap_model_set_residual_dtype(rap_model* m, int32_t dtype) {
  if (!m || (dtype != RAP_DT_F32 && dtype != RAP_DT_F16)) return RAP_ERR_INVALID;
  std::lock_guard<std::mutex> lock(m->cfg_mu);
  m->resid_dtype = dtype;
  return RAP_OK;
}
extern "C" int rap_model_residual_dtype(const rap_model* m) { return m ? m->resid_dtype : RAP_ERR_INVALID; }
extern "C" int rap_model_bounded_attention_launches(const rap_model* m) { return m ? (m->qk_norm ? m->n_bounded : 0) : RAP_ERR_INVALID; }
// qk_norm = False of the reference's constructor (point_cloud_dit.py:28, layer.py:75-83,103-104): q and k go to the attention as projected.
// Without the norm there is no bound on the logits, so every attention launch takes the online-softmax kernel; the gains in the blob are ignored.
extern "C" int rap_model_set_qk_norm(rap_model* m, int32_t on) {
  if (!m) return RAP_ERR_INVALID;
  std::lock_guard<std::mutex> lock(m->cfg_mu);
  m->qk_norm = on != 0;
  return RAP_OK;
}
extern "C" int rap_model_qk_norm(const rap_model* m) { return m ? (m->qk_norm ? 1 : 0) : RAP_ERR_INVALID; }

// ---------------------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------------------
// the arithmetic a call of `rows` token rows actually runs in (see g_rap_x2_min_rows)
static int eff_dtype(const rap_model* m, size_t rows) {
  return (m->dtype == RAP_DT_F32X2 && (long)rows < (long)g_rap_x2_min_rows) ? RAP_DT_F32 : m->dtype;
}

struct Workspace {
  float *base, *h, *xn, *qkv, *att, *ffmid, *ax, *v, *mod, *ada_scratch, *xt, *Rc, *tc, *tgrid;
  u16* h16;                            // the residual stream when it is held in fp16 (16-bit modes with resid_dtype = fp16; h is then null)
  float *hid1, *hid2, *astatic;        // head hidden layers (T,d), (T,d/2) and the static feature matrix (T,128): aliases
  u16 *xnh, *qkh, *vth, *atth, *ffmidh; // reduced-precision mode: 16-bit activations (xn/qkv/att/ffmid are then unused)
  float* splitk_h;                      // reduced-precision mode, few-token calls only: fp32 partial planes of the split-K ff2 GEMM (else null)
  int splitk_planes;                    // ... and how many (T, d) planes were reserved there (2 or 4; 0 without the buffer)
  int vt_nblk;
  int rows;                             // TQ = align_up(TP, 256): rows of every token-row buffer
  int dtype;                            // the arithmetic this call runs in (eff_dtype: the model's, or fp32 for a small split-precision call)
  bool qk_norm;                         // snapshot of the model's switch (the configuration mutex is held only while the layout is carved)
  double* proc_partials;
  int32_t *token_sample, *part_offsets, *attn_sort;
  int32_t *cu_batch_s, *cu_part_s;      // sanitised copies of the caller's segment tables (clamped to [0, TP], non-decreasing)
  const int32_t* cu_part_live;          // the part table the attention work lists were built from (part_offsets in rap_sample, cu_part_s in rap_dit_forward)
  int nseg_part, nseg_batch;            // segments in the two tables
  AttnWorkItem *items_batch, *items_part;
  int max_items_batch, max_items_part;
  int attn_bq;                          // query rows per work item of this call's attention lists
  int attn_kg;                          // key groups per block of the 16-bit attention (1, or 2 / 4 for few-token calls)
  size_t total;
};

static Workspace carve_workspace(const rap_model* m, int64_t TP, int B, int nseg_part, int rows, char* basep, int force_dtype = -1) {
  Workspace w;
  const size_t d = m->d, L = m->L;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* r = basep ? basep + off : nullptr; off += align_up(bytes, 256); return r; };
  // Every token-row buffer is carved at TQ = align_up(TP, 256) rows (round 4): the layer kernels run over TQ rows, so the persistent
  // 256-row-tile GEMMs serve ANY token count (ragged batches are the reference's real regime, RAP_inference.yaml:30-36); rows
  // TP .. TQ-1 are finite filler nobody reads (rows are independent in every kernel but attention, whose work lists and key
  // ranges stop at the true segment ends).  See forward_step.
  const size_t T = align_up((size_t)TP, 256);
  w.rows = (int)T;
  const int dtype = w.dtype = force_dtype >= 0 ? force_dtype : eff_dtype(m, T);
  w.qk_norm = m->qk_norm;
  w.base = (float*)take(T * d * 4);
  const bool x2 = dtype == RAP_DT_F32X2;    // split precision: fp32 residual stream, every 16-bit activation buffer holds heads AND tails
  const bool h16 = dtype != RAP_DT_F32 && !x2 && m->resid_dtype == RAP_DT_F16;
  w.h = h16 ? nullptr : (float*)take(T * d * 4);
  w.h16 = h16 ? (u16*)take(T * d * 2) : nullptr;
  w.xn = w.qkv = w.att = w.ffmid = nullptr;
  w.xnh = w.qkh = w.vth = w.atth = w.ffmidh = nullptr;
  w.splitk_h = nullptr;
  w.splitk_planes = 0;
  w.vt_nblk = 0;
  if (dtype == RAP_DT_F32) {
    w.xn = (float*)take(T * d * 4);            // also head hidden 1 (TP,d)
    w.qkv = (float*)take(T * 3 * d * 4);
    w.att = (float*)take(T * d * 4);           // also head hidden 2 (TP,d/2)
    w.ffmid = (float*)take(T * 4 * d * 4);     // also the static feature matrix (TP,128) during prepare
    w.hid1 = w.xn; w.hid2 = w.att; w.astatic = w.ffmid;
  } else {
    const size_t pl = x2 ? 2 : 1;              // planes per value
    w.vt_nblk = (int)(T / 64);                 // V^T image: whole 64-token blocks over the padded rows
    w.xnh = (u16*)take(T * d * 2 * pl);
    w.qkh = (u16*)take(T * 2 * d * 2 * pl);    // q,k [2][H][T][64]   (x2: [2][H][2 chunks][T][64 physical])
    w.vth = (u16*)take((size_t)w.vt_nblk * 64 * d * 2 * pl);   // [H][vt_nblk][64][64]   (x2: [H][vt_nblk][2 chunks][64][64 physical])
    w.atth = (u16*)take(T * d * 2 * pl);
    w.ffmidh = (u16*)take(T * 4 * d * 2 * pl); // >= 8*T*d bytes: also hosts the fp32 head hidden layers / static features
    w.hid1 = (float*)w.ffmidh;                 // (T,d) fp32   = 4*T*d bytes
    w.hid2 = w.hid1 + T * d;                   // (T,d/2) fp32 = 2*T*d bytes
    w.astatic = (float*)w.ffmidh;
    // ff2: the one layer GEMM with K >= 1024.  Reserved by SHAPE alone (tuning key 6 only gates the launch), so that the size
    // rap_workspace_bytes reports cannot change between the query and the call
    const int splits = gemm_h16_splits_by_shape((int)T, (int)d, (int)((x2 ? 8 : 4) * d));   // (split precision: the physical K of ff2 is 8 d)
    if (splits > 1) { w.splitk_h = (float*)take((size_t)splits * T * d * 4); w.splitk_planes = splits; }
  }
  w.ax = (float*)take(T * 64 * 4);
  w.v = (float*)take(T * 3 * 4);
  w.xt = (float*)take(T * 3 * 4);
  w.mod = (float*)take((size_t)rows * 2 * L * 2 * d * 4);
  w.ada_scratch = (float*)take((size_t)rows * (256 + 4 * L * d) * 4);
  w.tgrid = (float*)take((size_t)rows * 4);
  w.token_sample = (int32_t*)take(T * 4);
  w.part_offsets = (int32_t*)take(((size_t)nseg_part + 1) * 4);
  w.cu_batch_s = (int32_t*)take(((size_t)B + 1) * 4);
  w.cu_part_s = (int32_t*)take(((size_t)nseg_part + 1) * 4);
  w.cu_part_live = w.part_offsets; w.nseg_part = nseg_part; w.nseg_batch = B;
  // query rows per attention work item: 256, or 64 / 128 for few-token calls of the 16-bit modes (attention_h16_block_queries)
  w.attn_bq = (dtype == RAP_DT_BF16 || dtype == RAP_DT_F16) ? attention_h16_block_queries(dtype, (long)T) : RAP_ATTN_BQ;
  w.attn_kg = (dtype == RAP_DT_BF16 || dtype == RAP_DT_F16) ? attention_h16_key_groups(dtype, (long)T) : 1;
  w.max_items_batch = (int)(TP / w.attn_bq) + B + 1;
  w.max_items_part = (int)(TP / w.attn_bq) + nseg_part + 1;
  // (reserved for the smallest work items whatever tuning key 20 says: the size rap_workspace_bytes reports does not depend on the key)
  w.items_batch = (AttnWorkItem*)take(((size_t)(TP / 64) + B + 1) * sizeof(AttnWorkItem));
  w.items_part = (AttnWorkItem*)take(((size_t)(TP / 64) + nseg_part + 1) * sizeof(AttnWorkItem));
  w.attn_sort = (int32_t*)take(((size_t)(nseg_part > B ? nseg_part : B) + 1) * 4);   // scratch of the longest-first work-list order
  w.proc_partials = (double*)take((size_t)nseg_part * RAP_PROC_CHUNKS * 16 * 8);
  w.Rc = (float*)take((size_t)nseg_part * 9 * 4);
  w.tc = (float*)take((size_t)nseg_part * 3 * 4);
  w.total = off;
  return w;
}

extern "C" size_t rap_workspace_bytes(const rap_model* m, int64_t TP, int32_t B, int32_t nseg_part, int32_t rows) {
  if (!m || TP < 0 || B < 0 || nseg_part < 0 || rows < 0) return 0;
  std::lock_guard<std::mutex> lock(m->cfg_mu);
  // A split-precision model runs calls below tuning key 17's threshold on the exact-fp32 kernels, and the two layouts differ in size (the
  // split layout adds the split-K planes of few-token calls).  The size reported covers BOTH, so that a change of the process-global
  // threshold between this query and the call -- __graft_entry__.smoke() and the tests do toggle it -- cannot turn a valid call into
  // RAP_ERR_WORKSPACE (the comment here used to claim the two layouts need the same bytes).
  if (m->dtype == RAP_DT_F32X2) {
    const size_t a = carve_workspace(m, TP, B, nseg_part, rows, nullptr, RAP_DT_F32).total;
    const size_t b = carve_workspace(m, TP, B, nseg_part, rows, nullptr, RAP_DT_F32X2).total;
    return a > b ? a : b;
  }
  return carve_workspace(m, TP, B, nseg_part, rows, nullptr).total;
}

// ---------------------------------------------------------------------------------------------
// prepare (step-invariant work) and one forward
// ---------------------------------------------------------------------------------------------
static int zero_rows(hipStream_t stream, void* base, size_t row_bytes, int row0, int row1) {
  if (row1 <= row0) return RAP_OK;
  if (hipMemsetAsync((char*)base + (size_t)row0 * row_bytes, 0, (size_t)(row1 - row0) * row_bytes, stream) != hipSuccess) {
    rap_set_last_hip_error((int)hipGetLastError());
    return RAP_ERR_HIP;
  }
  return RAP_OK;
}

static int prepare_static(const rap_model* m, const Workspace& w, hipStream_t stream, const float* cond, const float* feat,
                          const float* scales, const uint8_t* anchor, const int32_t* cu_batch, const int32_t* cu_part,
                          int B, int nseg_part, int TP, const float* latent = nullptr) {
  int rc;
  // the caller's tables are only ever read through sanitised copies: every index derived from them stays inside [0, TP]
  if ((rc = launch_sanitize_cu(stream, cu_batch, B + 1, TP, w.cu_batch_s))) return rc;
  cu_batch = w.cu_batch_s;
  if (cu_part != w.part_offsets) {          // rap_dit_forward: the caller's part table (rap_sample builds its own, clamped, from points_per_part)
    if ((rc = launch_sanitize_cu(stream, cu_part, nseg_part + 1, TP, w.cu_part_s))) return rc;
    cu_part = w.cu_part_s;
  }
  if ((rc = launch_token_sample(stream, cu_batch, B, w.token_sample))) return rc;
  const int bq = w.dtype == RAP_DT_F32 ? 0 : w.dtype == RAP_DT_F32X2 ? 256 : w.attn_bq;
  int32_t* sort_ws = g_rap_attn_lpt ? w.attn_sort : nullptr;
  if ((rc = launch_build_attn_worklist(stream, cu_batch, B, w.items_batch, w.max_items_batch, bq, sort_ws))) return rc;
  if ((rc = launch_build_attn_worklist(stream, cu_part, nseg_part, w.items_part, w.max_items_part, bq, sort_ws))) return rc;
  // Filler rows TP .. TQ-1 (TQ = align_up(TP, 256), see carve_workspace): every value a layer kernel reads there has to be FINITE,
  // because the 16-bit attention's V^T image is blocked by 64 tokens and a masked key still multiplies its V column by p = 0
  // (0 x NaN = NaN).  Zero the filler rows of the four buffers no kernel writes beyond row TP -- the step-invariant embedding
  // part, PE(x_t), the attention output, the sample index of a token -- once per call; everything else in rows TP .. TQ-1 is then
  // computed from those by row-independent kernels (LayerNorm, GEMMs) and stays finite.
  const int TQ = w.rows, d = m->d;
  if ((rc = zero_rows(stream, w.base, (size_t)d * 4, TP, TQ))) return rc;
  if ((rc = zero_rows(stream, w.ax, 64 * 4, TP, TQ))) return rc;
  if ((rc = zero_rows(stream, w.token_sample, 4, TP, TQ))) return rc;
  if (w.dtype == RAP_DT_F32) { if ((rc = zero_rows(stream, w.att, (size_t)d * 4, TP, TQ))) return rc; }
  else if ((rc = zero_rows(stream, w.atth, (size_t)d * 2 * (w.dtype == RAP_DT_F32X2 ? 2 : 1), TP, TQ))) return rc;
  // base = [PE63(cond) | PE21(scale) | feat | 0] Wstatic^T + emb bias + anchor embedding   (embedding.py:155-179,
  // point_cloud_dit.py:119-139) -- everything in the embedding that does not depend on x_t.
  float* astatic = w.astatic;
  const int Ks = m->Ks;
  if ((rc = launch_posenc_static(stream, cond, scales, w.token_sample, feat, m->F, astatic, TP, Ks))) return rc;
  if (m->Din > 0) {
    // latent point features (embedding.py:163-166: concatenated between the coordinate and the scale embeddings): step-invariant, so
    // they are Din more columns of the hoisted K = Ks GEMM; columns 128 + Din .. Ks - 1 are zero padding up to the k-tile
    if (Ks > 128 + m->Din &&
        hipMemset2DAsync(astatic + 128 + m->Din, (size_t)Ks * 4, 0, (size_t)(Ks - 128 - m->Din) * 4, (size_t)TP, stream) != hipSuccess) {
      rap_set_last_hip_error((int)hipGetLastError());
      return RAP_ERR_HIP;
    }
    if ((rc = launch_copy_cols(stream, latent, m->Din, 0, astatic, Ks, 128, TP, m->Din))) return rc;
  }
  GemmParams g{};
  g.A = astatic; g.lda = Ks; g.W = m->Wstatic; g.ldw = Ks; g.C = w.base; g.ldc = m->d;
  g.M = TP; g.N = m->d; g.K = Ks; g.bias = m->emb_bias; g.anchor = anchor; g.anchor_emb = m->anchor_emb;
  return launch_gemm_f32(stream, EPI_BIAS_ANCHOR, g);
}

// mod: (scale|shift) rows for this forward: row r at mod + r*mod_stride, LN j at + j*2d.
static int forward_step(const rap_model* m, const Workspace& w, hipStream_t stream, const float* x_t, const float* mod,
                        long mod_stride, const int32_t* token_row, int TP_valid, float* v_out, float* feats_out) {
  const int d = m->d, H = m->H;
  int rc;
  // embed = PE63(x_t) Wx^T + base
  { ProfScope ps(stream, 4); rc = launch_posenc_x(stream, x_t, w.ax, TP_valid); }
  if (rc) return rc;
  // From here to the head every kernel runs over TQ = align_up(TP, 256) rows of the workspace (filler rows: prepare_static):
  // the GEMMs see M % 256 == 0 whatever the batch, i.e. the persistent 256 x 256 kernels; the attention launches take TQ as the
  // row count of the head-major planes and their work lists (true segment ends) for everything else.
  const int TP = w.rows;
  {
    GemmParams g{};
    // 16-bit residual stream: the fp32 embedding lands in the (idle) FFN buffer and is rounded to fp16 once
    g.A = w.ax; g.lda = 64; g.W = m->Wx; g.ldw = 64; g.C = w.h16 ? w.hid1 : w.h; g.ldc = d; g.M = TP; g.N = d; g.K = 64;
    g.resid = w.base; g.ldr = d;
    if ((rc = launch_gemm_f32(stream, EPI_BIAS_RESID, g))) return rc;
    if (w.h16 && (rc = launch_convert_f16_sat(stream, w.hid1, w.h16, (size_t)TP * d))) return rc;
  }
  const int dt = w.dtype;
  const void* hres = w.h16 ? (const void*)w.h16 : (const void*)w.h;      // the residual stream as the 16-bit LayerNorms read it
  const int hres_f16 = w.h16 ? 1 : 0;
  const int epi_resid = w.h16 ? EPI_H_BIAS_RESID_H16 : EPI_H_BIAS_RESID_F32;
  // Few-token calls of the 16-bit / split-precision blocks (round 6; the reference's everyday batch_size: 1, RAP_inference.yaml:30-36): every
  // residual GEMM leaves fp32 partial planes (split-K where gemm_h16_splits says so, one plane otherwise) and ONE kernel forms the new
  // residual-stream value and the LayerNorm that follows it (launch_resid_combine_ln_h16) -- 11 launches per layer instead of 14.  The planes
  // live in the split-K buffer, which exists exactly for the calls this is meant for (<= 128 tiles of 128 x 128 in the N = d GEMMs).
  const bool fused = g_rap_small_fused && dt != RAP_DT_F32 && w.splitk_h != nullptr && g_rap_gemm_splitk;
  auto fused_splits = [](const GemmParamsH& g) { const int sp = gemm_h16_splits(g.M, g.N, g.K); return sp > 1 ? sp : 1; };
  bool xn_ready = false;                    // the LayerNorm output the next QKV projection reads is already in xnh
  for (int i = 0; i < m->L; ++i) {
    const LayerW& lw = m->layers[i];
    if (dt == RAP_DT_F32X2) {
      // ---- split-precision block (round 5): fp32-accurate products on the fp16 matrix pipe.  Same kernel sequence as the 16-bit
      // block below on paired head / tail operands (physical K = 2 d resp. 8 d), fp32 residual stream, qk-norm always fused,
      // online softmax (probabilities must fit fp16).
      const LayerWH& lh = m->half[dt].layers[i];
      for (int a = 0; a < 2; ++a) {
        const int j = 2 * i + a;
        if (!xn_ready) { ProfScope ps(stream, 3); rc = launch_layernorm_mod_h16(stream, dt, w.h, 0, w.xnh, TP, d, mod + (size_t)j * 2 * d, mod_stride, token_row); }
        if (rc) return rc;
        GemmParamsH g{};
        g.A = w.xnh; g.lda = 2 * d; g.W = lh.Wqkv[a]; g.ldw = 2 * d; g.C = w.qkh; g.M = TP; g.N = 3 * d; g.K = 2 * d; g.heads = H;
        g.vt = w.vth; g.vt_nblk = w.vt_nblk; g.q_mul = 8.0f; g.acc_scale = lh.s_qkv[a]; g.out_scale = lh.g_v[a];
        if (w.qk_norm) { g.gamma_q = lw.gq[a]; g.gamma_k = lw.gk[a]; }      // null gains: the epilogue splits q / k as projected
        { ProfScope ps(stream, 2); rc = launch_gemm_h16(stream, dt, EPI_H_QKV_NORM, g); }
        if (rc) return rc;
        {
          ProfScope ps(stream, a);
          // few-token calls: the keys of every work item over 2 / 4 blocks; the partial O planes (4 x TQ x d floats) live in the idle FFN
          // buffer (16 TQ d bytes in this mode), the partial (max, row sum) pairs in the idle LayerNorm-output buffer
          const int max_items = a == 0 ? w.max_items_part : w.max_items_batch;
          const int splits = attention_x2_splits(max_items, H);
          rc = launch_attention_x2(stream, w.qkh, w.vth, w.vt_nblk, w.atth, TP, H, a == 0 ? w.items_part : w.items_batch, max_items,
                                   reinterpret_cast<float*>(w.ffmidh), reinterpret_cast<float*>(w.xnh), splits, TP_valid,
                                   a == 0 ? w.cu_part_live : w.cu_batch_s, a == 0 ? w.nseg_part : w.nseg_batch);
        }
        if (rc) return rc;
        GemmParamsH o{};
        o.A = w.atth; o.lda = 2 * d; o.W = lh.Wout[a]; o.ldw = 2 * d; o.C = w.h; o.ldc = d; o.M = TP; o.N = d; o.K = 2 * d;
        o.bias = lw.bout[a]; o.resid = w.h; o.ldr = d; o.acc_scale = lh.s_out[a] / lh.g_v[a];      // (the attention output carries V's activation scale)
        if (fused) { o.splitk_ws = w.splitk_h; o.defer_combine = 1; o.force_splits = fused_splits(o); }      // (the count is fixed HERE: the GEMM and the pass that reads its planes cannot disagree)
        { ProfScope ps(stream, 2); rc = launch_gemm_h16(stream, dt, EPI_H_BIAS_RESID_F32, o); }
        if (rc) return rc;
        if (fused) {      // combine pass + the NEXT LayerNorm in one kernel: adaLN of the per-sample branch after a = 0, the FFN's affine LN after a = 1
          ProfScope ps(stream, 3);
          rc = launch_resid_combine_ln_h16(stream, dt, w.splitk_h, o.force_splits, o.bias, w.h, 0, w.xnh, TP, d,
                                           a == 0 ? mod + (size_t)(j + 1) * 2 * d : nullptr, mod_stride, token_row, lw.ffn_g, lw.ffn_b);
          if (rc) return rc;
          xn_ready = true;
        }
      }
      if (!fused) { ProfScope ps(stream, 3); rc = launch_layernorm_affine_h16(stream, dt, w.h, 0, w.xnh, TP, d, lw.ffn_g, lw.ffn_b); }
      if (rc) return rc;
      GemmParamsH f1{};
      f1.A = w.xnh; f1.lda = 2 * d; f1.W = lh.Wff1p; f1.ldw = 2 * d; f1.C = w.ffmidh; f1.ldc = 8 * d; f1.M = TP; f1.N = 8 * d; f1.K = 2 * d;
      f1.bias = lw.bff1p; f1.acc_scale = lh.s_ff1; f1.out_scale = lh.g_ff1;
      { ProfScope ps(stream, 2); rc = launch_gemm_h16(stream, dt, EPI_H_GEGLU, f1); }
      if (rc) return rc;
      GemmParamsH f2{};
      f2.A = w.ffmidh; f2.lda = 8 * d; f2.W = lh.Wff2; f2.ldw = 8 * d; f2.C = w.h; f2.ldc = d; f2.M = TP; f2.N = d; f2.K = 8 * d;
      f2.bias = lw.bff2; f2.resid = w.h; f2.ldr = d; f2.acc_scale = lh.s_ff2 / lh.g_ff1;
      f2.splitk_ws = w.splitk_h;               // few-token calls: the physical K = 8d split over 2 / 4 blocks per tile (null otherwise)
      const bool f2_fused = fused && i + 1 < m->L;      // ... and its combine pass is the first LayerNorm of the next layer
      if (f2_fused) { f2.defer_combine = 1; f2.force_splits = fused_splits(f2); }
      { ProfScope ps(stream, 2); rc = launch_gemm_h16(stream, dt, EPI_H_BIAS_RESID_F32, f2); }
      if (rc) return rc;
      xn_ready = false;
      if (f2_fused) {
        ProfScope ps(stream, 3);
        rc = launch_resid_combine_ln_h16(stream, dt, w.splitk_h, f2.force_splits, f2.bias, w.h, 0, w.xnh, TP, d,
                                         mod + (size_t)(2 * i + 2) * 2 * d, mod_stride, token_row, nullptr, nullptr);
        if (rc) return rc;
        xn_ready = true;
      }
      continue;
    }
    if (dt != RAP_DT_F32) {
      // ---- reduced-precision block: 16-bit MFMA operands, fp32 accumulate, fp32 residual stream / LN / softmax
      const LayerWH& lh = m->half[dt].layers[i];
      for (int a = 0; a < 2; ++a) {
        const int j = 2 * i + a;
        if (!xn_ready) { ProfScope ps(stream, 3); rc = launch_layernorm_mod_h16(stream, dt, hres, hres_f16, w.xnh, TP, d, mod + (size_t)j * 2 * d, mod_stride, token_row); }
        if (rc) return rc;
        GemmParamsH g{};
        g.A = w.xnh; g.lda = d; g.W = lh.Wqkv[a]; g.ldw = d; g.C = w.qkh; g.M = TP; g.N = 3 * d; g.K = d; g.heads = H;
        g.vt = w.vth; g.vt_nblk = w.vt_nblk;
        const bool bnd = w.qk_norm && m->bounded[j] != 0;                // this launch's softmax kernel (per layer and branch)
        const bool prescale = attention_h16_wants_prescaled_q(dt, bnd);
        // (few-token calls: the fused epilogue exists on 128 x 128 tiles too since round 6 -- launch_gemm_h16 picks the tile; until then
        // such calls ran the projection and qk-norm as two kernels, r03 call 32)
        if (g_rap_fuse_qknorm && w.qk_norm) {
          // qk-norm in the QKV epilogue: one kernel, q / k normalised from the fp32 accumulators (no 16-bit round trip through HBM)
          g.gamma_q = lw.gq[a]; g.gamma_k = lw.gk[a]; g.q_mul = prescale ? RAP_QMUL_PRESCALED : 8.0f;
          { ProfScope ps(stream, 2); rc = launch_gemm_h16(stream, dt, EPI_H_QKV_NORM, g); }
          if (rc) return rc;
        } else {
          { ProfScope ps(stream, 2); rc = launch_gemm_h16(stream, dt, EPI_H_QKV, g); }
          if (rc) return rc;
          if (w.qk_norm && (rc = launch_qknorm_h16(stream, dt, w.qkh, TP, H, lw.gq[a], lw.gk[a], prescale ? RAP_QMUL_PRESCALED : 8.0f))) return rc;
        }
        {
          ProfScope ps(stream, a);
          const float* bound = bnd ? m->logit_bound + (size_t)j * H : nullptr;
          rc = launch_attention_h16(stream, dt, w.qkh, w.vth, w.vt_nblk, w.atth, TP, H, a == 0 ? w.items_part : w.items_batch,
                                    a == 0 ? w.max_items_part : w.max_items_batch, bound, prescale ? 1 : 0, w.attn_bq, w.attn_kg);
        }
        if (rc) return rc;
        GemmParamsH o{};
        o.A = w.atth; o.lda = d; o.W = lh.Wout[a]; o.ldw = d; o.ldc = d; o.M = TP; o.N = d; o.K = d;
        o.bias = lw.bout[a]; o.ldr = d;
        if (w.h16) { o.C = w.h16; o.resid_h = w.h16; } else { o.C = w.h; o.resid = w.h; }
        if (fused) { o.splitk_ws = w.splitk_h; o.defer_combine = 1; o.force_splits = fused_splits(o); }
        { ProfScope ps(stream, 2); rc = launch_gemm_h16(stream, dt, epi_resid, o); }
        if (rc) return rc;
        if (fused) {      // combine pass + the NEXT LayerNorm in one kernel (see the split-precision block)
          ProfScope ps(stream, 3);
          rc = launch_resid_combine_ln_h16(stream, dt, w.splitk_h, o.force_splits, o.bias, w.h16 ? (void*)w.h16 : (void*)w.h, hres_f16, w.xnh, TP, d,
                                           a == 0 ? mod + (size_t)(j + 1) * 2 * d : nullptr, mod_stride, token_row, lw.ffn_g, lw.ffn_b);
          if (rc) return rc;
          xn_ready = true;
        }
      }
      if (!fused) { ProfScope ps(stream, 3); rc = launch_layernorm_affine_h16(stream, dt, hres, hres_f16, w.xnh, TP, d, lw.ffn_g, lw.ffn_b); }
      if (rc) return rc;
      GemmParamsH f1{};
      f1.A = w.xnh; f1.lda = d; f1.W = lh.Wff1p; f1.ldw = d; f1.C = w.ffmidh; f1.ldc = 4 * d; f1.M = TP; f1.N = 8 * d; f1.K = d;
      f1.bias = lw.bff1p;
      { ProfScope ps(stream, 2); rc = launch_gemm_h16(stream, dt, EPI_H_GEGLU, f1); }
      if (rc) return rc;
      GemmParamsH f2{};
      f2.A = w.ffmidh; f2.lda = 4 * d; f2.W = lh.Wff2; f2.ldw = 4 * d; f2.ldc = d; f2.M = TP; f2.N = d; f2.K = 4 * d;
      f2.bias = lw.bff2; f2.ldr = d;
      f2.splitk_ws = w.splitk_h;               // few-token calls: K = 4d split over 2 / 4 blocks per tile (null otherwise)
      if (w.h16) { f2.C = w.h16; f2.resid_h = w.h16; } else { f2.C = w.h; f2.resid = w.h; }
      const bool f2_fused = fused && i + 1 < m->L;
      if (f2_fused) { f2.defer_combine = 1; f2.force_splits = fused_splits(f2); }
      { ProfScope ps(stream, 2); rc = launch_gemm_h16(stream, dt, epi_resid, f2); }
      if (rc) return rc;
      xn_ready = false;
      if (f2_fused) {
        ProfScope ps(stream, 3);
        rc = launch_resid_combine_ln_h16(stream, dt, w.splitk_h, f2.force_splits, f2.bias, w.h16 ? (void*)w.h16 : (void*)w.h, hres_f16, w.xnh, TP, d,
                                         mod + (size_t)(2 * i + 2) * 2 * d, mod_stride, token_row, nullptr, nullptr);
        if (rc) return rc;
        xn_ready = true;
      }
      continue;
    }
    for (int a = 0; a < 2; ++a) {   // a = 0: per-part attention, a = 1: per-sample attention (layer.py:152-160)
      const int j = 2 * i + a;
      { ProfScope ps(stream, 3); rc = launch_layernorm_mod(stream, w.h, w.xn, TP, d, mod + (size_t)j * 2 * d, mod_stride, token_row); }
      if (rc) return rc;
      GemmParams g{};
      g.A = w.xn; g.lda = d; g.W = lw.Wqkv[a]; g.ldw = d; g.C = w.qkv; g.M = TP; g.N = 3 * d; g.K = d; g.heads = H;
      const bool fuse_qk = g_rap_fuse_qknorm != 0 && w.qk_norm;
      if (fuse_qk) { g.gamma_q = lw.gq[a]; g.gamma_k = lw.gk[a]; }        // qk-norm in the QKV epilogue (tuning key 7)
      { ProfScope ps(stream, 2); rc = launch_gemm_f32(stream, EPI_QKV_HEADMAJOR, g); }
      if (rc) return rc;
      if (!fuse_qk && w.qk_norm && (rc = launch_qknorm(stream, w.qkv, TP, H, lw.gq[a], lw.gk[a]))) return rc;
      {
        ProfScope ps(stream, a);
        const float* bound = (w.qk_norm && m->bounded[j]) ? m->logit_bound + (size_t)j * H : nullptr;
        // few-token calls: split the keys of every work item over up to 4 blocks; the partial O planes live in the (idle) FFN
        // buffer (4 x TP x d floats), the partial row sums in the (idle) LN-output buffer
        const int max_items = a == 0 ? w.max_items_part : w.max_items_batch;
        const int splits = attention_f32_splits(max_items, H, bound != nullptr);
        rc = launch_attention_f32(stream, w.qkv, w.att, TP, H, a == 0 ? w.items_part : w.items_batch, max_items, bound, w.ffmid, w.xn,
                                  splits, a == 0 ? w.cu_part_live : w.cu_batch_s, a == 0 ? w.nseg_part : w.nseg_batch);
      }
      if (rc) return rc;
      GemmParams o{};
      o.A = w.att; o.lda = d; o.W = lw.Wout[a]; o.ldw = d; o.C = w.h; o.ldc = d; o.M = TP; o.N = d; o.K = d;
      o.bias = lw.bout[a]; o.resid = w.h; o.ldr = d;
      o.splitk_ws = w.ffmid;     // idle between the FFNs (the split attention's partial planes are consumed by now): 4 partial (T,d) planes
      { ProfScope ps(stream, 2); rc = launch_gemm_f32(stream, EPI_BIAS_RESID, o); }
      if (rc) return rc;
    }
    { ProfScope ps(stream, 3); rc = launch_layernorm_affine(stream, w.h, w.xn, TP, d, lw.ffn_g, lw.ffn_b); }
    if (rc) return rc;
    GemmParams f1{};
    f1.A = w.xn; f1.lda = d; f1.W = lw.Wff1p; f1.ldw = d; f1.C = w.ffmid; f1.ldc = 4 * d; f1.M = TP; f1.N = 8 * d; f1.K = d;
    f1.bias = lw.bff1p;
    { ProfScope ps(stream, 2); rc = launch_gemm_f32(stream, EPI_GEGLU, f1); }
    if (rc) return rc;
    GemmParams f2{};
    f2.A = w.ffmid; f2.lda = 4 * d; f2.W = lw.Wff2; f2.ldw = 4 * d; f2.C = w.h; f2.ldc = d; f2.M = TP; f2.N = d; f2.K = 4 * d;
    f2.bias = lw.bff2; f2.resid = w.h; f2.ldr = d;
    f2.splitk_ws = w.qkv;      // idle during the FFN; qkv (3 T d) and att (T d) are adjacent: 4 partial (T,d) planes
    { ProfScope ps(stream, 2); rc = launch_gemm_f32(stream, EPI_BIAS_RESID, f2); }
    if (rc) return rc;
  }
  // 16-bit residual stream: the fp32 head (and the caller's transformer_features) read an fp32 image of it -- written into the
  // caller's feature buffer when there is one, else into the q / k planes (T x 2d 16-bit values = T x d floats, dead after the last attention)
  const float* h_final = w.h;
  int head_rows = TP;                       // the head's first GEMM reads TQ rows of a workspace buffer, TP_valid rows of a caller's
  if (w.h16) {
    float* img = feats_out ? feats_out : reinterpret_cast<float*>(w.qkh);
    if (feats_out) head_rows = TP_valid;
    if ((rc = launch_convert_f16_to_f32(stream, w.h16, img, (size_t)head_rows * d))) return rc;
    h_final = img;
  } else if (feats_out) {
    if (hipMemcpyAsync(feats_out, w.h, (size_t)TP_valid * d * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess) {
      rap_set_last_hip_error((int)hipGetLastError());
      return RAP_ERR_HIP;
    }
  }
  // final_mlp (point_cloud_dit.py:111-117): Lin+SiLU, Lin+SiLU, Lin(no bias)
  GemmParams h0{};
  h0.A = h_final; h0.lda = d; h0.W = m->hW0; h0.ldw = d; h0.C = w.hid1; h0.ldc = d; h0.M = head_rows; h0.N = d; h0.K = d; h0.bias = m->hb0;
  // few-token calls: K of the two hidden layers over 2 / 4 blocks per tile (launch_gemm_f32 decides by the tile count); the partial planes
  // live in the FFN buffer of the fp32 layout (4 T d floats, idle here) resp. the split-K planes of the 16-bit layouts
  float* const head_ws = w.dtype == RAP_DT_F32 ? w.ffmid : w.splitk_h;
  const int head_planes = w.dtype == RAP_DT_F32 ? 4 : w.splitk_planes;
  h0.splitk_ws = head_ws; h0.splitk_planes = head_planes;
  if ((rc = launch_gemm_f32(stream, EPI_BIAS_SILU, h0))) return rc;
  GemmParams h2{};
  h2.A = w.hid1; h2.lda = d; h2.W = m->hW2; h2.ldw = d; h2.C = w.hid2; h2.ldc = d / 2; h2.M = head_rows; h2.N = d / 2; h2.K = d;
  h2.bias = m->hb2; h2.splitk_ws = head_ws; h2.splitk_planes = head_planes;
  if ((rc = launch_gemm_f32(stream, EPI_BIAS_SILU, h2))) return rc;
  return launch_head_out3(stream, w.hid2, d / 2, m->hW4, v_out, TP_valid, d / 2);
}

extern "C" int rap_dit_forward(const rap_model* m, const float* x_t, const float* timesteps, const float* cond,
                               const float* feat, const float* scales, const uint8_t* anchor, const int32_t* cu_batch,
                               const int32_t* cu_part, int32_t B, int32_t VP, int64_t TP, float* v_out, float* feats_out,
                               void* ws, size_t ws_bytes, void* stream_) {
  return rap_dit_forward_latent(m, x_t, timesteps, cond, feat, nullptr, scales, anchor, cu_batch, cu_part, B, VP, TP, v_out, feats_out, ws, ws_bytes, stream_);
}
extern "C" int rap_dit_forward_latent(const rap_model* m, const float* x_t, const float* timesteps, const float* cond,
                                      const float* feat, const float* latent, const float* scales, const uint8_t* anchor, const int32_t* cu_batch,
                                      const int32_t* cu_part, int32_t B, int32_t VP, int64_t TP, float* v_out, float* feats_out,
                                      void* ws, size_t ws_bytes, void* stream_) {
  if (!m || !x_t || !timesteps || !cond || !scales || !anchor || !cu_batch || !cu_part || !v_out) return RAP_ERR_INVALID;
  if (m->F > 0 && !feat) return RAP_ERR_INVALID;
  if ((m->Din > 0) != (latent != nullptr)) return RAP_ERR_INVALID;      // a model built with in_dim > 0 needs its latent features, one without takes none
  if (B <= 0 || VP < 0 || TP < 0 || TP > 0x7fffffffLL / 8) return RAP_ERR_INVALID;
  if (TP == 0) return RAP_OK;
  if (!ws) return RAP_ERR_WORKSPACE;
  // the configuration (compute / residual dtype, qk_norm) is snapshotted into the workspace descriptor under the model's mutex; the enqueue
  // itself runs unlocked, so threads sharing a model do not serialise on a (possibly queue-bound) enqueue
  Workspace w;
  { std::lock_guard<std::mutex> lock(m->cfg_mu); w = carve_workspace(m, TP, B, VP, B, (char*)ws); }
  if (w.total > ws_bytes) return RAP_ERR_WORKSPACE;
  w.cu_part_live = w.cu_part_s;             // prepare_static sanitises the caller's part table into it
  hipStream_t stream = (hipStream_t)stream_;
  int rc;
  if ((rc = prepare_static(m, w, stream, cond, feat, scales, anchor, cu_batch, cu_part, B, VP, (int)TP, latent))) return rc;
  if ((rc = launch_adaln_table(stream, timesteps, B, 2 * m->L, m->d, m->adaW1, m->adab1, m->adaW2, m->adab2, m->adaW3,
                               m->adab3, w.ada_scratch, w.mod)))
    return rc;
  return forward_step(m, w, stream, x_t, w.mod, (long)2 * m->L * 2 * m->d, w.token_sample, (int)TP, v_out, feats_out);
}

extern "C" int rap_euler_step(const float* x_t, const float* v, float t, float dt, float* x0_hat, float* x_next,
                              float* traj_xt_slot, int64_t n, void* stream) {
  if (!x_t || !v || !x0_hat || !x_next || n < 0) return RAP_ERR_INVALID;
  return launch_euler_step((hipStream_t)stream, x_t, v, t, dt, x0_hat, x_next, traj_xt_slot, (long)n);
}

// ---------------------------------------------------------------------------------------------
// procrustes entry points
// ---------------------------------------------------------------------------------------------
struct ProcWs { int32_t* off; double* partials; float* Rc; float* tc; size_t total; };
static ProcWs carve_proc(int nparts, char* basep) {
  ProcWs p; size_t off = 0;
  auto take = [&](size_t bytes) { char* r = basep ? basep + off : nullptr; off += align_up(bytes, 256); return r; };
  p.off = (int32_t*)take(((size_t)nparts + 1) * 4);
  p.partials = (double*)take((size_t)nparts * RAP_PROC_CHUNKS * 16 * 8);
  p.Rc = (float*)take((size_t)nparts * 9 * 4);
  p.tc = (float*)take((size_t)nparts * 3 * 4);
  p.total = off;
  return p;
}
extern "C" size_t rap_procrustes_workspace_bytes(int32_t nparts) { return nparts < 0 ? 0 : carve_proc(nparts, nullptr).total; }

extern "C" int rap_fit_transformations(const float* src, const float* tgt, const int64_t* points_per_part, int32_t B,
                                       int32_t P, float* R_out, float* t_out, void* ws, size_t ws_bytes, void* stream_) {
  if (!src || !tgt || !points_per_part || !R_out || !t_out || B <= 0 || P <= 0) return RAP_ERR_INVALID;
  if ((int64_t)B * P > 65535) return RAP_ERR_INVALID;
  const int np = B * P;
  if (!ws) return RAP_ERR_WORKSPACE;
  ProcWs p = carve_proc(np, (char*)ws);
  if (p.total > ws_bytes) return RAP_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  int rc;
  if ((rc = launch_part_offsets(stream, points_per_part, np, p.off))) return rc;
  return launch_procrustes_fit(stream, src, tgt, p.off, np, R_out, t_out, p.partials);
}

extern "C" int rap_rigidify(const float* prediction, const float* condition, const int64_t* points_per_part, int32_t B,
                            int32_t P, float* out, void* ws, size_t ws_bytes, void* stream_) {
  if (!prediction || !condition || !points_per_part || !out || B <= 0 || P <= 0) return RAP_ERR_INVALID;
  if ((int64_t)B * P > 65535) return RAP_ERR_INVALID;
  const int np = B * P;
  if (!ws) return RAP_ERR_WORKSPACE;
  ProcWs p = carve_proc(np, (char*)ws);
  if (p.total > ws_bytes) return RAP_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  int rc;
  if ((rc = launch_part_offsets(stream, points_per_part, np, p.off))) return rc;
  if ((rc = launch_procrustes_fit(stream, condition, prediction, p.off, np, p.Rc, p.tc, p.partials))) return rc;
  return launch_rigid_apply(stream, condition, p.Rc, p.tc, p.off, np, out, nullptr, 0.f, 0.f, nullptr, 0);
}

extern "C" int rap_rigidify_blend(const float* x0_hat, const float* condition, const int64_t* points_per_part, int32_t B,
                                  int32_t P, const float* x_1, float w0, float w1, float* x_t_out, void* ws, size_t ws_bytes,
                                  void* stream_) {
  if (!x0_hat || !condition || !points_per_part || !x_1 || !x_t_out || B <= 0 || P <= 0) return RAP_ERR_INVALID;
  if ((int64_t)B * P > 65535) return RAP_ERR_INVALID;
  const int np = B * P;
  if (!ws) return RAP_ERR_WORKSPACE;
  ProcWs p = carve_proc(np, (char*)ws);
  if (p.total > ws_bytes) return RAP_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  int rc;
  if ((rc = launch_part_offsets(stream, points_per_part, np, p.off))) return rc;
  if ((rc = launch_procrustes_fit(stream, condition, x0_hat, p.off, np, p.Rc, p.tc, p.partials))) return rc;
  return launch_rigid_apply(stream, condition, p.Rc, p.tc, p.off, np, x_t_out, x_1, w0, w1, nullptr, 1);
}

// ---------------------------------------------------------------------------------------------
// the whole sampling loop
// ---------------------------------------------------------------------------------------------
__global__ void tgrid_kernel(float* t, int steps, double dt) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s < steps) t[s] = (float)(1.0 - (double)s * dt);   // t = 1 - step*dt in double (sampler.py:42,55), cast at torch.full
}

extern "C" int rap_sample(const rap_model* m, const float* cond, const float* feat, const float* scales,
                          const uint8_t* anchor, const int64_t* points_per_part, const int32_t* cu_batch, const float* x_1,
                          int32_t B, int32_t P, int64_t TP, int32_t num_steps, int32_t rigidity_forcing, float* traj_x0,
                          float* traj_xt, float* R_out, float* t_out, float* feats_out, void* ws, size_t ws_bytes,
                          void* stream_) {
  return rap_sample_latent(m, cond, feat, nullptr, scales, anchor, points_per_part, cu_batch, x_1, B, P, TP, num_steps, rigidity_forcing, traj_x0,
                           traj_xt, R_out, t_out, feats_out, ws, ws_bytes, stream_);
}
extern "C" int rap_sample_latent(const rap_model* m, const float* cond, const float* feat, const float* latent, const float* scales,
                                 const uint8_t* anchor, const int64_t* points_per_part, const int32_t* cu_batch, const float* x_1,
                                 int32_t B, int32_t P, int64_t TP, int32_t num_steps, int32_t rigidity_forcing, float* traj_x0,
                                 float* traj_xt, float* R_out, float* t_out, float* feats_out, void* ws, size_t ws_bytes,
                                 void* stream_) {
  if (!m || !cond || !scales || !anchor || !points_per_part || !cu_batch || !x_1 || !traj_x0 || !traj_xt || !R_out || !t_out)
    return RAP_ERR_INVALID;
  if (m->F > 0 && !feat) return RAP_ERR_INVALID;
  if ((m->Din > 0) != (latent != nullptr)) return RAP_ERR_INVALID;
  if (B <= 0 || P <= 0 || TP <= 0 || num_steps <= 0 || TP > 0x7fffffffLL / 8 || (int64_t)B * P > 65535) return RAP_ERR_INVALID;
  const int np = B * P;
  if (!ws) return RAP_ERR_WORKSPACE;
  Workspace w;
  { std::lock_guard<std::mutex> lock(m->cfg_mu); w = carve_workspace(m, TP, B, np, num_steps, (char*)ws); }      // see rap_dit_forward
  if (w.total > ws_bytes) return RAP_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int T = (int)TP;
  const long n3 = (long)TP * 3;
  int rc;
  if ((rc = launch_part_offsets(stream, points_per_part, np, w.part_offsets, (long)TP))) return rc;
  if ((rc = prepare_static(m, w, stream, cond, feat, scales, anchor, cu_batch, w.part_offsets, B, np, T, latent))) return rc;
  // t is uniform over the batch inside the sampler (modeling.py:674), so the adaLN table is computed once for
  // ALL flow steps (row s = step s) instead of per sample per step.
  const double dt = 1.0 / (double)num_steps;
  hipLaunchKernelGGL(tgrid_kernel, dim3((num_steps + 63) / 64), dim3(64), 0, stream, w.tgrid, num_steps, dt);
  RAP_LAUNCH_CHECK();
  if ((rc = launch_adaln_table(stream, w.tgrid, num_steps, 2 * m->L, m->d, m->adaW1, m->adab1, m->adaW2, m->adab2,
                               m->adaW3, m->adab3, w.ada_scratch, w.mod)))
    return rc;
  RAP_HIP_CHECK(hipMemcpyAsync(w.xt, x_1, (size_t)n3 * sizeof(float), hipMemcpyDeviceToDevice, stream));
  const size_t mod_step = (size_t)2 * m->L * 2 * m->d;
  for (int s = 0; s < num_steps; ++s) {
    const double t = 1.0 - (double)s * dt;
    float* x0_slot = traj_x0 + (size_t)s * n3;
    float* xt_slot = traj_xt + (size_t)s * n3;
    float* feats = (feats_out && s == num_steps - 1) ? feats_out : nullptr;
    if ((rc = forward_step(m, w, stream, w.xt, w.mod + (size_t)s * mod_step, 0, nullptr, T, w.v, feats))) return rc;
    { ProfScope ps(stream, 5); rc = launch_euler_step(stream, w.xt, w.v, (float)t, (float)dt, x0_slot, w.xt, rigidity_forcing ? nullptr : xt_slot, n3); }
    if (rc) return rc;
    if (rigidity_forcing) {
      { ProfScope ps(stream, 6); rc = launch_procrustes_fit(stream, cond, x0_slot, w.part_offsets, np, w.Rc, w.tc, w.proc_partials); }
      if (rc) return rc;
      const float w0 = (float)(1.0 - t + dt), w1 = (float)(t - dt);
      { ProfScope ps(stream, 7); rc = launch_rigid_apply(stream, cond, w.Rc, w.tc, w.part_offsets, np, w.xt, x_1, w0, w1, xt_slot, 1); }
      if (rc) return rc;
    }
  }
  // final poses (modeling.py:389-391): fit_transformations(cond, trajs[-1])
  return launch_procrustes_fit(stream, cond, traj_x0 + (size_t)(num_steps - 1) * n3, w.part_offsets, np, R_out, t_out,
                               w.proc_partials);
}

// ---------------------------------------------------------------------------------------------
// kernel-level entry points
// ---------------------------------------------------------------------------------------------
extern "C" int rap_gemm_f32(int32_t epilogue, const float* A, int32_t lda, const float* W, int32_t ldw, float* C, int32_t ldc,
                            int32_t M, int32_t N, int32_t K, const float* bias, const float* resid, int32_t ldr,
                            const uint8_t* anchor, const float* anchor_emb, int32_t heads, void* stream) {
  if (!A || !W || !C) return RAP_ERR_INVALID;
  if (epilogue == EPI_BIAS_RESID && !resid) return RAP_ERR_INVALID;
  if (epilogue == EPI_BIAS_ANCHOR && (!anchor || !anchor_emb)) return RAP_ERR_INVALID;
  GemmParams g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias;
  g.resid = resid; g.ldr = ldr; g.anchor = anchor; g.anchor_emb = anchor_emb; g.heads = heads;
  return launch_gemm_f32((hipStream_t)stream, epilogue, g);
}

extern "C" int rap_geglu_interleave(const float* W, const float* b, float* Wp, float* bp, int32_t inner, int32_t K,
                                    void* stream) {
  if (!W || !b || !Wp || !bp) return RAP_ERR_INVALID;
  return launch_geglu_interleave((hipStream_t)stream, W, b, Wp, bp, inner, K);
}

// [work items | sanitised copy of the caller's cu_seqlens]: the kernel-level attention entry points index with the copy only (ADVICE r05:
// an entry above TP in the caller's table made seg_start + q run past the planes; the model path has sanitised its tables since round 5)
// (work items reserved at the smallest granularity, 64 rows: tuning key 20's forced item sizes reach rap_attention_h16 too)
static size_t attn_items_bytes(int64_t TP, int32_t nseg) { return align_up(((size_t)(TP / 64) + (size_t)nseg + 1) * sizeof(AttnWorkItem), 256); }
extern "C" size_t rap_attention_workspace_bytes(int64_t TP, int32_t nseg) {
  return attn_items_bytes(TP, nseg) + align_up(((size_t)nseg + 1) * sizeof(int32_t), 256);
}
// -> device pointer to the clamped, non-decreasing copy of cu_seqlens inside ws
static int attn_ws_sanitize(hipStream_t stream, const int32_t* cu_seqlens, int32_t nseg, int64_t TP, void* ws, const int32_t** cu_out) {
  int32_t* cu_s = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + attn_items_bytes(TP, nseg));
  *cu_out = cu_s;
  return launch_sanitize_cu(stream, cu_seqlens, nseg + 1, (long)TP, cu_s);
}

// the work list of one attention launch as rap_sample / rap_dit_forward build it: one {seg_start, seg_len, q0, 0} item per
// `block_queries` query rows of every segment, longest segment first when sort_ws (nseg ints) is given, zero items after the last
extern "C" int rap_build_attention_worklist(const int32_t* cu_seqlens, int32_t nseg, int32_t block_queries, int32_t* items_out,
                                            int32_t max_items, int32_t* sort_ws, void* stream) {
  if (!cu_seqlens || !items_out || nseg < 0 || max_items < 0 || block_queries <= 0) return RAP_ERR_INVALID;
  return launch_build_attn_worklist((hipStream_t)stream, cu_seqlens, nseg, reinterpret_cast<AttnWorkItem*>(items_out), max_items,
                                    block_queries, sort_ws);
}

extern "C" int rap_attention_f32(const float* qkv_headmajor, const int32_t* cu_seqlens, int32_t nseg, float* out,
                                 int64_t TP, int32_t heads, const float* logit_bound, void* ws, size_t ws_bytes,
                                 void* stream_) {
  if (!qkv_headmajor || !cu_seqlens || !out || nseg < 0 || TP < 0 || TP > 0x7fffffffLL / 8) return RAP_ERR_INVALID;
  if (!ws || ws_bytes < rap_attention_workspace_bytes(TP, nseg)) return RAP_ERR_WORKSPACE;
  const int max_items = (int)(TP / RAP_ATTN_BQ) + nseg + 1;
  hipStream_t stream = (hipStream_t)stream_;
  int rc;
  if ((rc = attn_ws_sanitize(stream, cu_seqlens, nseg, TP, ws, &cu_seqlens))) return rc;
  if ((rc = launch_build_attn_worklist(stream, cu_seqlens, nseg, (AttnWorkItem*)ws, max_items, 0))) return rc;
  return launch_attention_f32(stream, qkv_headmajor, out, (int)TP, heads, (const AttnWorkItem*)ws, max_items, logit_bound, nullptr, nullptr, 1);
}

extern "C" int rap_layernorm_mod(const float* x, float* out, int64_t TP, int32_t d, const float* mod, int64_t mod_stride,
                                 const int32_t* token_row, void* stream) {
  if (!x || !out || !mod) return RAP_ERR_INVALID;
  return launch_layernorm_mod((hipStream_t)stream, x, out, (int)TP, d, mod, (long)mod_stride, token_row);
}
extern "C" int rap_layernorm_affine(const float* x, float* out, int64_t TP, int32_t d, const float* gain, const float* shift,
                                    void* stream) {
  if (!x || !out || !gain || !shift) return RAP_ERR_INVALID;
  return launch_layernorm_affine((hipStream_t)stream, x, out, (int)TP, d, gain, shift);
}
extern "C" int rap_qknorm(float* qkv_headmajor, int64_t TP, int32_t heads, const float* gamma_q, const float* gamma_k,
                          void* stream) {
  if (!qkv_headmajor || !gamma_q || !gamma_k) return RAP_ERR_INVALID;
  return launch_qknorm((hipStream_t)stream, qkv_headmajor, (int)TP, heads, gamma_q, gamma_k);
}
extern "C" int rap_posenc_x(const float* x, float* ax, int64_t TP, void* stream) {
  if (!x || !ax) return RAP_ERR_INVALID;
  return launch_posenc_x((hipStream_t)stream, x, ax, (int)TP);
}
extern "C" int rap_posenc_static(const float* cond, const float* scales, const int32_t* token_sample, const float* feat,
                                 int32_t feat_dim, float* astatic, int64_t TP, void* stream) {
  if (!cond || !scales || !token_sample || !astatic) return RAP_ERR_INVALID;
  return launch_posenc_static((hipStream_t)stream, cond, scales, token_sample, feat, feat_dim, astatic, (int)TP);
}
extern "C" int rap_token_sample(const int32_t* cu_batch, int32_t B, int32_t* token_sample, void* stream) {
  if (!cu_batch || !token_sample) return RAP_ERR_INVALID;
  return launch_token_sample((hipStream_t)stream, cu_batch, B, token_sample);
}
extern "C" int rap_adaln_table(const rap_model* m, const float* t, int32_t rows, float* scratch, float* out, void* stream) {
  if (!m || !t || !scratch || !out) return RAP_ERR_INVALID;
  return launch_adaln_table((hipStream_t)stream, t, rows, 2 * m->L, m->d, m->adaW1, m->adab1, m->adaW2, m->adab2, m->adaW3,
                            m->adab3, scratch, out);
}

// ---- reduced-precision kernel-level entry points (dtype: 1 = bf16, 2 = fp16; 16-bit tensors as uint16_t*) ----
extern "C" int rap_convert_h16(int32_t dtype, const float* src, uint16_t* dst, int64_t n, void* stream) {
  if (!src || !dst || n < 0) return RAP_ERR_INVALID;
  return launch_convert_h16((hipStream_t)stream, dtype, src, dst, (size_t)n);
}
extern "C" int rap_gemm_h16(int32_t dtype, int32_t epilogue, const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw,
                            void* C, int32_t ldc, int32_t M, int32_t N, int32_t K, const float* bias, const float* resid,
                            int32_t ldr, int32_t heads, uint16_t* vt, int32_t vt_nblk, void* stream) {
  if (!A || !W || !C) return RAP_ERR_INVALID;
  GemmParamsH g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias;
  g.resid = resid; g.ldr = ldr; g.heads = heads; g.vt = vt; g.vt_nblk = vt_nblk;
  if (epilogue == EPI_H_BIAS_RESID_H16) { g.resid_h = reinterpret_cast<const uint16_t*>(resid); g.resid = nullptr; }   // fp16 residual
  return launch_gemm_h16((hipStream_t)stream, dtype, epilogue, g);
}
// the residual epilogues (1, 6) with an optional split-K workspace (what rap_sample hands ff2 on few-token calls)
extern "C" size_t rap_gemm_h16_splitk_workspace_bytes(int32_t M, int32_t N, int32_t K) {
  const int s = gemm_h16_splits_by_shape(M, N, K);
  return s > 1 ? (size_t)s * (size_t)M * (size_t)N * sizeof(float) : 0;
}
extern "C" int rap_gemm_h16_splitk(int32_t dtype, int32_t epilogue, const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, void* C,
                                   int32_t ldc, int32_t M, int32_t N, int32_t K, const float* bias, const void* resid, int32_t ldr,
                                   void* ws, size_t ws_bytes, void* stream) {
  if (!A || !W || !C || (epilogue != EPI_H_BIAS_RESID_F32 && epilogue != EPI_H_BIAS_RESID_H16)) return RAP_ERR_INVALID;
  const size_t need = rap_gemm_h16_splitk_workspace_bytes(M, N, K);
  if (need && (!ws || ws_bytes < need)) return RAP_ERR_WORKSPACE;
  GemmParamsH g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias; g.ldr = ldr;
  if (epilogue == EPI_H_BIAS_RESID_H16) g.resid_h = reinterpret_cast<const uint16_t*>(resid); else g.resid = reinterpret_cast<const float*>(resid);
  g.splitk_ws = need ? reinterpret_cast<float*>(ws) : nullptr;
  return launch_gemm_h16((hipStream_t)stream, dtype, epilogue, g);
}
extern "C" int rap_attention_h16(int32_t dtype, const uint16_t* qk, const uint16_t* vt, int32_t vt_nblk,
                                 const int32_t* cu_seqlens, int32_t nseg, uint16_t* out, int64_t TP, int32_t heads,
                                 const float* logit_bound, void* ws, size_t ws_bytes, void* stream_) {
  if (!qk || !vt || !cu_seqlens || !out || nseg < 0 || TP < 0 || TP > 0x7fffffffLL / 8) return RAP_ERR_INVALID;
  if (!ws || ws_bytes < rap_attention_workspace_bytes(TP, nseg)) return RAP_ERR_WORKSPACE;
  // 256-row work items, two stages -- unless tuning key 20 FORCES an item size / key groups (64, 128, 66, 130: the A/B values), which
  // then applies here as it does to a model call of the same row count (so the few-token kernels have kernel-level tests)
  const long rows = attention_h16_forced() ? (long)align_up((size_t)TP, 256) : 0;
  const int bq = attention_h16_block_queries(dtype, rows), kg = attention_h16_key_groups(dtype, rows);
  const int max_items = (int)(TP / bq) + nseg + 1;
  hipStream_t stream = (hipStream_t)stream_;
  int rc;
  if ((rc = attn_ws_sanitize(stream, cu_seqlens, nseg, TP, ws, &cu_seqlens))) return rc;
  if ((rc = launch_build_attn_worklist(stream, cu_seqlens, nseg, (AttnWorkItem*)ws, max_items, bq))) return rc;
  return launch_attention_h16(stream, dtype, qk, vt, vt_nblk, out, (int)TP, heads, (const AttnWorkItem*)ws, max_items, logit_bound, 0, bq, kg);
}
extern "C" int rap_layernorm_mod_h16(int32_t dtype, const float* x, uint16_t* out, int64_t TP, int32_t d, const float* mod,
                                     int64_t mod_stride, const int32_t* token_row, void* stream) {
  if (!x || !out || !mod) return RAP_ERR_INVALID;
  return launch_layernorm_mod_h16((hipStream_t)stream, dtype, x, 0, out, (int)TP, d, mod, (long)mod_stride, token_row);
}
extern "C" int rap_layernorm_affine_h16(int32_t dtype, const float* x, uint16_t* out, int64_t TP, int32_t d, const float* gain,
                                        const float* shift, void* stream) {
  if (!x || !out || !gain || !shift) return RAP_ERR_INVALID;
  return launch_layernorm_affine_h16((hipStream_t)stream, dtype, x, 0, out, (int)TP, d, gain, shift);
}
extern "C" int rap_qknorm_h16(int32_t dtype, uint16_t* qk, int64_t TP, int32_t heads, const float* gamma_q,
                              const float* gamma_k, void* stream) {
  if (!qk || !gamma_q || !gamma_k) return RAP_ERR_INVALID;
  return launch_qknorm_h16((hipStream_t)stream, dtype, qk, (int)TP, heads, gamma_q, gamma_k, 8.0f);
}

// ---- split-precision kernel-level entry points (compute dtype 3; paired fp16 head / tail operands, see include/rapflow.h) ----
extern "C" int rap_x2_pack(const float* src, int64_t ld_src, int64_t rows, int32_t cols, float scale, uint16_t* dst, void* stream) {
  if (!src || !dst || rows < 0 || cols < 0 || ld_src < cols) return RAP_ERR_INVALID;
  return launch_x2_pack((hipStream_t)stream, src, (long)ld_src, (long)rows, cols, scale, dst);
}
extern "C" int rap_x2_unpack(const uint16_t* src, int64_t rows, int32_t cols, float inv_scale, float* dst, void* stream) {
  if (!src || !dst || rows < 0 || cols < 0) return RAP_ERR_INVALID;
  return launch_x2_unpack((hipStream_t)stream, src, (long)rows, cols, inv_scale, dst);
}
extern "C" int rap_x2_gemm(int32_t epilogue, const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, void* C, int32_t ldc, int32_t M,
                           int32_t N, int32_t K_physical, const float* bias, const float* resid, int32_t ldr, float acc_scale, int32_t heads,
                           const float* gamma_q, const float* gamma_k, float q_mul, uint16_t* vt, int32_t vt_nblk, void* stream) {
  if (!A || !W || !C) return RAP_ERR_INVALID;
  GemmParamsH g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K_physical; g.bias = bias;
  g.resid = resid; g.ldr = ldr; g.heads = heads; g.vt = vt; g.vt_nblk = vt_nblk; g.gamma_q = gamma_q; g.gamma_k = gamma_k;
  g.q_mul = q_mul; g.acc_scale = acc_scale;
  return launch_gemm_h16((hipStream_t)stream, RAP_DT_F32X2, epilogue, g);
}
extern "C" int rap_x2_attention(const uint16_t* qk, const uint16_t* vt, int32_t vt_nblk, const int32_t* cu_seqlens, int32_t nseg,
                                uint16_t* out, int64_t TP, int32_t heads, void* ws, size_t ws_bytes, void* stream_) {
  if (!qk || !vt || !cu_seqlens || !out || nseg < 0 || TP < 0 || TP > 0x7fffffffLL / 16) return RAP_ERR_INVALID;
  if (!ws || ws_bytes < rap_attention_workspace_bytes(TP, nseg)) return RAP_ERR_WORKSPACE;
  const int max_items = (int)(TP / RAP_ATTN_BQ) + nseg + 1;
  hipStream_t stream = (hipStream_t)stream_;
  int rc;
  if ((rc = attn_ws_sanitize(stream, cu_seqlens, nseg, TP, ws, &cu_seqlens))) return rc;
  if ((rc = launch_build_attn_worklist(stream, cu_seqlens, nseg, (AttnWorkItem*)ws, max_items, 256))) return rc;
  return launch_attention_x2(stream, qk, vt, vt_nblk, out, (int)TP, heads, (const AttnWorkItem*)ws, max_items);
}

// ---------------------------------------------------------------------------------------------
// generation selection by rigidity
// ---------------------------------------------------------------------------------------------
struct RigWs { int32_t* off; double* partials; float* Rc; float* tc; float* per_step; size_t total; };
static RigWs carve_rig(int nparts, int S, int B, char* basep) {
  RigWs p; size_t off = 0;
  auto take = [&](size_t bytes) { char* r = basep ? basep + off : nullptr; off += align_up(bytes, 256); return r; };
  p.off = (int32_t*)take(((size_t)nparts + 1) * 4);
  p.partials = (double*)take((size_t)nparts * RAP_PROC_CHUNKS * 16 * 8);
  p.Rc = (float*)take((size_t)nparts * 9 * 4);
  p.tc = (float*)take((size_t)nparts * 3 * 4);
  p.per_step = (float*)take((size_t)(S > 0 ? S : 1) * B * 4);
  p.total = off;
  return p;
}
extern "C" size_t rap_rigidity_workspace_bytes(int32_t nparts, int32_t steps, int32_t B) {
  return (nparts < 0 || steps < 0 || B < 0) ? 0 : carve_rig(nparts, steps, B, nullptr).total;
}

extern "C" int rap_rigidity_rmse(const float* cond, const float* pred, const float* R, const float* t,
                                 const int64_t* points_per_part, int32_t B, int32_t P, const float* scales,
                                 int32_t average_per_part, float* out, void* ws, size_t ws_bytes, void* stream_) {
  if (!cond || !pred || !R || !t || !points_per_part || !out || B <= 0 || P <= 0) return RAP_ERR_INVALID;
  if ((int64_t)B * P > 65535) return RAP_ERR_INVALID;
  if (!ws) return RAP_ERR_WORKSPACE;
  RigWs w = carve_rig(B * P, 0, B, (char*)ws);
  if (w.total > ws_bytes) return RAP_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  int rc;
  if ((rc = launch_part_offsets(stream, points_per_part, B * P, w.off))) return rc;
  return launch_rigidity_rmse(stream, cond, pred, R, t, w.off, B, P, scales, average_per_part, out, w.partials);
}

extern "C" int rap_trajectory_rigidity_rmse(const float* cond, const float* traj, const int64_t* points_per_part, int32_t B,
                                            int32_t P, int64_t TP, int32_t steps, const float* scales, float* mean_out,
                                            float* per_step_out, void* ws, size_t ws_bytes, void* stream_) {
  if (!cond || !traj || !points_per_part || !mean_out || B <= 0 || P <= 0 || TP <= 0 || steps <= 0) return RAP_ERR_INVALID;
  if ((int64_t)B * P > 65535) return RAP_ERR_INVALID;
  if (!ws) return RAP_ERR_WORKSPACE;
  RigWs w = carve_rig(B * P, steps, B, (char*)ws);
  if (w.total > ws_bytes) return RAP_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  float* per_step = per_step_out ? per_step_out : w.per_step;
  int rc;
  if ((rc = launch_part_offsets(stream, points_per_part, B * P, w.off))) return rc;
  for (int s = 0; s < steps; ++s) {
    const float* x0 = traj + (size_t)s * TP * 3;
    if ((rc = launch_procrustes_fit(stream, cond, x0, w.off, B * P, w.Rc, w.tc, w.partials))) return rc;      // modeling.py:479-481
    if ((rc = launch_rigidity_rmse(stream, cond, x0, w.Rc, w.tc, w.off, B, P, scales, 0, per_step + (size_t)s * B, w.partials)))
      return rc;                                                                                              // :484-487
  }
  return launch_step_mean(stream, per_step, steps, B, mean_out);                                             // :489
}

extern "C" int rap_select_generation(const float* rmse, int32_t G, int32_t B, int32_t P, int64_t TP, const int32_t* cu_batch,
                                     const float* clouds, const float* R, const float* t, int32_t pick_largest,
                                     int32_t* best_out, float* cloud_out, float* R_out, float* t_out, void* stream) {
  if (!rmse || !best_out || G <= 0 || B <= 0) return RAP_ERR_INVALID;
  if (clouds && (!R || !t || !cu_batch || !cloud_out || !R_out || !t_out || P <= 0 || TP <= 0)) return RAP_ERR_INVALID;
  return launch_select_generation((hipStream_t)stream, rmse, G, B, P, (long)TP, cu_batch, clouds, R, t, pick_largest, best_out,
                                  cloud_out, R_out, t_out);
}

extern "C" int rap_relative_transforms(const float* R_pred, const float* t_pred, const float* R_gt, const float* t_gt,
                                       const float* scales, const int64_t* points_per_part, int32_t B, int32_t P,
                                       const float* global_rotation, const float* global_translation, float* out, void* stream) {
  if (!R_pred || !t_pred || !R_gt || !t_gt || !scales || !points_per_part || !out || B <= 0 || P <= 0) return RAP_ERR_INVALID;
  if ((global_rotation == nullptr) != (global_translation == nullptr)) return RAP_ERR_INVALID;
  return launch_relative_transforms((hipStream_t)stream, R_pred, t_pred, R_gt, t_gt, scales, points_per_part, B, P,
                                    global_rotation, global_translation, out);
}

// ---------------------------------------------------------------------------------------------
// cross-part overlap ratio
// ---------------------------------------------------------------------------------------------
// compute_transform_errors without ICP (reference eval/metrics.py:165-303): see transforms.hip
extern "C" int rap_transform_errors(const float* R_gt, const float* t_gt, const float* R_pred, const float* t_pred,
                                    const int64_t* points_per_part, const uint8_t* anchor_part, const int64_t* matched_part_ids,
                                    const float* scale, int32_t B, int32_t P, float* rot_err_per_part, float* trans_err_per_part,
                                    float* rot_err_mean, float* trans_err_mean, void* stream) {
  if (!R_gt || !t_gt || !R_pred || !t_pred || !points_per_part || !anchor_part || !rot_err_per_part || !trans_err_per_part ||
      !rot_err_mean || !trans_err_mean || B <= 0 || P <= 0)
    return RAP_ERR_INVALID;
  return launch_transform_errors((hipStream_t)stream, R_gt, t_gt, R_pred, t_pred, points_per_part, anchor_part, matched_part_ids, scale,
                                 B, P, rot_err_per_part, trans_err_per_part, rot_err_mean, trans_err_mean);
}

struct OvWs { int32_t* off; int32_t* pid; float* min_dist; void* items; size_t total; };
static OvWs carve_ov(int64_t TP, int B, int P, char* basep) {
  OvWs w; size_t off = 0;
  auto take = [&](size_t bytes) { char* r = basep ? basep + off : nullptr; off += align_up(bytes, 256); return r; };
  w.off = (int32_t*)take(((size_t)B * P + 1) * 4);
  w.pid = (int32_t*)take((size_t)TP * 4);
  w.min_dist = (float*)take((size_t)TP * 4);
  w.items = take(overlap_max_items((long)TP, B) * 16);
  w.total = off;
  return w;
}
extern "C" size_t rap_overlap_workspace_bytes(int64_t TP, int32_t B, int32_t P) {
  return (TP < 0 || B < 0 || P < 0) ? 0 : carve_ov(TP, B, P, nullptr).total;
}
extern "C" int rap_overlap_ratio(const float* pointclouds_pred, const int64_t* points_per_part, const int32_t* cu_batch, int32_t B,
                                 int32_t P, int64_t TP, const float* h_taus, int32_t n_taus, float* ratios_out, float* min_dist_out,
                                 void* ws, size_t ws_bytes, void* stream_) {
  if (!pointclouds_pred || !points_per_part || !cu_batch || !h_taus || !ratios_out || B <= 0 || P <= 0 || TP <= 0) return RAP_ERR_INVALID;
  if ((int64_t)B * P > 65535 || TP > 0x7fffffffLL / 8) return RAP_ERR_INVALID;
  if (!ws) return RAP_ERR_WORKSPACE;
  OvWs w = carve_ov(TP, B, P, (char*)ws);
  if (w.total > ws_bytes) return RAP_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  int rc;
  if ((rc = launch_part_offsets(stream, points_per_part, B * P, w.off))) return rc;
  return launch_overlap_ratio(stream, pointclouds_pred, cu_batch, w.off, B, P, (long)TP, h_taus, n_taus, ratios_out,
                              min_dist_out ? min_dist_out : w.min_dist, w.pid, w.items);
}

// ---------------------------------------------------------------------------------------------
// MiniSpinNet local feature extractor
// ---------------------------------------------------------------------------------------------
#include <cmath>
struct SpinLayer { int Cin, Cout, Kd, ldw; bool bn_relu; const float* W; const float* b; const float* Wt; const float* bt; };   // Wt / bt: tap-major (implicit GEMM)
struct rap_spinnet {
  float* raw = nullptr;       // the caller's blob (MiniSpinNet.state_dict() float tensors, registration order)
  float* derived = nullptr;   // folded + padded conv weights / biases, voxel table, pool weights
  float h_w1[48], h_b1[16];   // point MLP with BatchNorm folded (kernel argument)
  const float* vox;           // (420,3)
  const void* pool_w;         // SpinPoolW on the device
  SpinLayer layers[8];
  const float* zeros = nullptr;   // 256 bytes of zeros: the elevation pad of the implicit-GEMM convolutions
};
static const int kSpinCin[8] = {16, 64, 64, 128, 128, 64, 64, 32};
static const int kSpinCout[8] = {64, 64, 128, 128, 64, 64, 32, 32};

extern "C" int64_t rap_spinnet_weight_count(void) {
  int64_t n = 48 + 16 * 5 + (512 + 16 * 5 + 16 + 5);
  for (int i = 0; i < 8; ++i) {
    const int64_t kd = (int64_t)kSpinCin[i] * (i == 0 ? 27 : 9);
    n += kSpinCout[i] * kd + kSpinCout[i] + (i < 7 ? 2 * kSpinCout[i] : 0);
  }
  return n;
}

extern "C" void rap_spinnet_destroy(rap_spinnet* m) {
  if (!m) return;
  if (m->raw) (void)hipFree(m->raw);
  if (m->derived) (void)hipFree(m->derived);
  delete m;
}

extern "C" int rap_spinnet_create(const float* d_weights, int64_t n_floats, void* stream_, rap_spinnet** out) {
  if (!out) return RAP_ERR_INVALID;
  *out = nullptr;
  if (!d_weights || n_floats != rap_spinnet_weight_count()) return RAP_ERR_INVALID;
  hipStream_t stream = (hipStream_t)stream_;
  rap_spinnet* m = new (std::nothrow) rap_spinnet();
  if (!m) return RAP_ERR_ALLOC;
  auto fail = [&](int code) { rap_spinnet_destroy(m); return code; };
  if (hipMalloc((void**)&m->raw, (size_t)n_floats * 4) != hipSuccess) { delete m; return RAP_ERR_ALLOC; }
  size_t n_der = 420 * 3 + 1024;      // voxel table + pool struct (613 floats used)
  for (int i = 0; i < 8; ++i) n_der += (size_t)128 * (i == 0 ? 448 : kSpinCin[i] * 9) + 128 + (i ? (size_t)kSpinCout[i] * kSpinCin[i] * 9 + 128 : (size_t)64 * 448 + 128);
  n_der += 64;      // zero page
  if (hipMalloc((void**)&m->derived, n_der * 4) != hipSuccess) return fail(RAP_ERR_ALLOC);
  if (hipMemcpyAsync(m->raw, d_weights, (size_t)n_floats * 4, hipMemcpyDeviceToDevice, stream) != hipSuccess) return fail(RAP_ERR_HIP);
  // ---- small heads on the host: point MLP (128 floats) and attention pool (613 floats)
  std::vector<float> h(128 + 613);
  if (hipMemcpyAsync(h.data(), m->raw, h.size() * 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return fail(RAP_ERR_HIP);
  if (hipStreamSynchronize(stream) != hipSuccess) return fail(RAP_ERR_HIP);
  {
    const float *W = &h[0], *b = &h[48], *g = &h[64], *be = &h[80], *rm = &h[96], *rv = &h[112];
    for (int c = 0; c < 16; ++c) {
      const float s = g[c] / std::sqrt(rv[c] + 1e-5f);
      for (int d = 0; d < 3; ++d) m->h_w1[c * 3 + d] = W[c * 3 + d] * s;
      m->h_b1[c] = (b[c] - rm[c]) * s + be[c];
    }
  }
  std::vector<float> pool(16 * 32 + 16 + 16 + 1);
  {
    const float* p0 = &h[128];
    const float *W1 = p0, *b1 = p0 + 512, *g1 = p0 + 528, *be1 = p0 + 544, *rm1 = p0 + 560, *rv1 = p0 + 576;
    const float *W2 = p0 + 592, *b2 = p0 + 608, *g2 = p0 + 609, *be2 = p0 + 610, *rm2 = p0 + 611, *rv2 = p0 + 612;
    for (int j = 0; j < 16; ++j) {
      const float s = g1[j] / std::sqrt(rv1[j] + 1e-5f);
      for (int c = 0; c < 32; ++c) pool[j * 32 + c] = W1[j * 32 + c] * s;
      pool[512 + j] = (b1[j] - rm1[j]) * s + be1[j];
    }
    const float s2 = g2[0] / std::sqrt(rv2[0] + 1e-5f);
    for (int j = 0; j < 16; ++j) pool[528 + j] = W2[j] * s2;
    pool[544] = (b2[0] - rm2[0]) * s2 + be2[0];
  }
  // ---- voxel centres: get_voxel_coordinate(radius 1, rad_n 3, azi_n 20, ele_n 7)  (utils/common.py:213-225, 338-372, 387-393)
  std::vector<float> vox(420 * 3);
  for (int r = 0; r < 3; ++r)
    for (int e = 0; e < 7; ++e)
      for (int a = 0; a < 20; ++a) {
        const double beta = M_PI * e / 7.0 + M_PI / 7.0 / 2.0, alpha = 2.0 * M_PI * a / 20.0 + M_PI / 20.0;
        const double sc = (double)r / 3.0 + 1.0 / 6.0;
        float* v = &vox[((r * 7 + e) * 20 + a) * 3];
        v[0] = (float)(sc * std::sin(beta) * std::cos(alpha));
        v[1] = (float)(sc * std::sin(beta) * std::sin(alpha));
        v[2] = (float)(sc * std::cos(beta));
      }
  float* q = m->derived;
  if (hipMemcpyAsync(q, vox.data(), vox.size() * 4, hipMemcpyHostToDevice, stream) != hipSuccess) return fail(RAP_ERR_HIP);
  m->vox = q; q += 420 * 3;
  if (hipMemcpyAsync(q, pool.data(), pool.size() * 4, hipMemcpyHostToDevice, stream) != hipSuccess) return fail(RAP_ERR_HIP);
  m->pool_w = q; q += 1024;
  if (hipStreamSynchronize(stream) != hipSuccess) return fail(RAP_ERR_HIP);   // vox / pool are stack-backed host vectors
  // ---- conv stack: fold BatchNorm (affine=False) and pad to 128 output columns / 32-multiple k
  const float* p = m->raw + 128 + 613;
  for (int i = 0; i < 8; ++i) {
    SpinLayer& L = m->layers[i];
    L.Cin = kSpinCin[i]; L.Cout = kSpinCout[i]; L.Kd = L.Cin * (i == 0 ? 27 : 9); L.ldw = i == 0 ? 448 : L.Kd; L.bn_relu = i < 7;
    const float* W = p; p += (size_t)L.Cout * L.Kd;
    const float* b = p; p += L.Cout;
    const float *rm = nullptr, *rv = nullptr;
    if (L.bn_relu) { rm = p; p += L.Cout; rv = p; p += L.Cout; }
    float* Wd = q; q += (size_t)128 * L.ldw;
    float* bd = q; q += 128;
    int rc;
    if ((rc = launch_spin_fold(stream, W, b, nullptr, nullptr, rm, rv, L.Cout, L.Kd, Wd, L.ldw, 128, bd))) return fail(rc);
    L.W = Wd; L.b = bd; L.Wt = nullptr; L.bt = nullptr;
    {
      const int ldt = i == 0 ? 448 : L.Kd;
      float* Wt = q; q += (size_t)L.Cout * ldt;
      float* bt = q; q += 128;
      if ((rc = launch_spin_fold_tapmajor(stream, W, b, rm, rv, L.Cin, L.Cout, i == 0 ? 27 : 9, ldt, Wt, bt))) return fail(rc);
      L.Wt = Wt; L.bt = bt;
    }
  }
  if (hipMemsetAsync(q, 0, 256, stream) != hipSuccess) return fail(RAP_ERR_HIP);
  m->zeros = q; q += 64;
  if ((int64_t)(p - m->raw) != n_floats) return fail(RAP_ERR_INVALID);
  *out = m;
  return RAP_OK;
}

struct SpinWs { float *x0, *A, *Y0, *Y1; size_t total; };
static SpinWs carve_spin(int Kc, char* basep) {
  SpinWs w; size_t off = 0;
  auto take = [&](size_t bytes) { char* r = basep ? basep + off : nullptr; off += align_up(bytes, 256); return r; };
  w.x0 = (float*)take((size_t)Kc * 420 * 16 * 4);
  w.A = (float*)take((size_t)Kc * 140 * 1152 * 4);
  w.Y0 = (float*)take((size_t)Kc * 140 * 128 * 4);
  w.Y1 = (float*)take((size_t)Kc * 140 * 128 * 4);
  w.total = off;
  return w;
}
extern "C" size_t rap_spinnet_workspace_bytes(int32_t keypoints_per_chunk) {
  return keypoints_per_chunk <= 0 ? 0 : carve_spin(keypoints_per_chunk, nullptr).total;
}

extern "C" int rap_spinnet_describe(const rap_spinnet* m, const float* pts, const int32_t* perm, int64_t N, const float* kpts,
                                    int32_t K, float des_r, int32_t flags, float* desc_out, int32_t keypoints_per_chunk, void* ws,
                                    size_t ws_bytes, void* stream_) {
  if (!m || !pts || !kpts || !desc_out || N <= 0 || K < 0 || !(des_r > 0.f) || keypoints_per_chunk <= 0) return RAP_ERR_INVALID;
  if (flags & ~(RAP_SPINNET_PATCH_LRF | RAP_SPINNET_IM2COL_PATH)) return RAP_ERR_INVALID;
  // per-call options (round 3, ADVICE r02: they used to be mutable state of the handle, a race between two callers of one model)
  const int lrf = (flags & RAP_SPINNET_PATCH_LRF) ? 1 : 0;
  const bool implicit = !(flags & RAP_SPINNET_IM2COL_PATH);
  if (K == 0) return RAP_OK;
  if (!ws) return RAP_ERR_WORKSPACE;
  SpinWs w = carve_spin(keypoints_per_chunk, (char*)ws);
  if (w.total > ws_bytes) return RAP_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  int rc;
  for (int k0 = 0; k0 < K; k0 += keypoints_per_chunk) {
    const int Kc = K - k0 < keypoints_per_chunk ? K - k0 : keypoints_per_chunk;
    const int M = Kc * 140;
    if ((rc = launch_spin_patch(stream, pts, perm, (long)N, kpts + (size_t)k0 * 3, Kc, des_r, m->vox, m->h_w1, m->h_b1, w.x0, lrf))) return rc;
    float* yin = nullptr;
    float* yout = w.Y0;
    int ld_in = 128;
    for (int i = 0; i < 8; ++i) {
      const SpinLayer& L = m->layers[i];
      if (implicit) {
        // implicit GEMM: the gather of the 3x3(x3) cylindrical neighbourhood happens in the DMA source addresses, output is dense (M, Cout)
        if (i == 0) rc = launch_spin_conv3d(stream, w.x0, L.Wt, L.bt, m->zeros, yout, M);
        else rc = launch_spin_conv3x3(stream, yin, ld_in, L.Cin, L.Wt, L.bt, m->zeros, yout, L.Cout, M, L.bn_relu);
        if (rc) return rc;
        ld_in = L.Cout;
      } else {
        if (i == 0) rc = launch_spin_im2col3d(stream, w.x0, Kc, w.A, L.ldw);
        else rc = launch_spin_im2col2d(stream, yin, 128, L.Cin, Kc, w.A);
        if (rc) return rc;
        GemmParams g{};
        g.A = w.A; g.lda = L.ldw; g.W = L.W; g.ldw = L.ldw; g.C = yout; g.ldc = 128; g.M = M; g.N = 128; g.K = L.ldw; g.bias = L.b;
        if ((rc = launch_gemm_f32(stream, L.bn_relu ? EPI_BIAS_RELU : EPI_BIAS, g))) return rc;
        ld_in = 128;
      }
      yin = yout; yout = (yout == w.Y0) ? w.Y1 : w.Y0;
    }
    if ((rc = launch_spin_pool(stream, yin, ld_in, Kc, m->pool_w, desc_out + (size_t)k0 * 32))) return rc;
  }
  return RAP_OK;
}

// ---------------------------------------------------------------------------------------------
// nearest-neighbour registration metrics
// ---------------------------------------------------------------------------------------------
struct NnWs { float* d2a; float* d2b; int32_t* nn; NnWork* items; size_t total; };
static NnWs carve_nn(int64_t n, int B, char* basep) {
  NnWs w; size_t off = 0;
  auto take = [&](size_t bytes) { char* r = basep ? basep + off : nullptr; off += align_up(bytes, 256); return r; };
  w.d2a = (float*)take((size_t)n * 4);
  w.d2b = (float*)take((size_t)n * 4);
  w.nn = (int32_t*)take((size_t)n * 4);
  w.items = (NnWork*)take(nn_max_items((long)n, B) * sizeof(NnWork));
  w.total = off;
  return w;
}
extern "C" size_t rap_nn_metrics_workspace_bytes(int64_t n_points, int32_t B) {
  return (n_points < 0 || B < 0) ? 0 : carve_nn(n_points, B, nullptr).total;
}
extern "C" int rap_chamfer_rmse(const float* pointclouds_gt, const float* pointclouds_pred, const int32_t* cu_batch, int32_t B,
                                int64_t TP, float* out, void* ws, size_t ws_bytes, void* stream) {
  if (!pointclouds_gt || !pointclouds_pred || !cu_batch || !out || B <= 0 || TP <= 0 || TP > 0x7fffffffLL / 8) return RAP_ERR_INVALID;
  if (!ws) return RAP_ERR_WORKSPACE;
  NnWs w = carve_nn(TP, B, (char*)ws);
  if (w.total > ws_bytes) return RAP_ERR_WORKSPACE;
  return launch_chamfer_rmse((hipStream_t)stream, pointclouds_gt, pointclouds_pred, cu_batch, B, (long)TP, out, w.d2a, w.d2b, w.items);
}
extern "C" int rap_correspondence_rmse(const float* source_gt, const float* target_gt, const float* source_pred,
                                       const float* target_pred, int32_t n_source, int32_t n_target, float distance_threshold,
                                       float* out3, void* ws, size_t ws_bytes, void* stream) {
  if (!source_gt || !target_gt || !source_pred || !target_pred || !out3 || n_source <= 0 || n_target <= 0) return RAP_ERR_INVALID;
  if (!ws) return RAP_ERR_WORKSPACE;
  NnWs w = carve_nn(n_source, 1, (char*)ws);
  if (w.total > ws_bytes) return RAP_ERR_WORKSPACE;
  return launch_correspondence_rmse((hipStream_t)stream, source_gt, target_gt, source_pred, target_pred, n_source, n_target,
                                    distance_threshold, out3, w.d2a, w.nn, w.items);
}

extern "C" int rap_farthest_point_sampling(const float* points, const int32_t* cloud_start, const int32_t* cloud_len,
                                           const int32_t* k_per_cloud, const int32_t* start_idx, int32_t n_clouds, int32_t k_max,
                                           int32_t* indices_out, float* dist_ws, void* stream) {
  if (!points || !cloud_start || !cloud_len || !k_per_cloud || !start_idx || !indices_out || !dist_ws || n_clouds <= 0 || k_max <= 0)
    return RAP_ERR_INVALID;
  return launch_fps((hipStream_t)stream, points, cloud_start, cloud_len, k_per_cloud, start_idx, n_clouds, k_max, dist_ws, indices_out);
}

// ---------------------------------------------------------------------------------------------
// voxel down-sampling (preprocessing)
// ---------------------------------------------------------------------------------------------
extern "C" int rap_voxel_bounds(const float* points, int64_t N, float voxel_size, int64_t* bounds6_out, float* dist_max_out,
                                void* stream) {
  if (!points || !bounds6_out || !dist_max_out || N <= 0 || !(voxel_size > 0.f)) return RAP_ERR_INVALID;
  return launch_voxel_bounds((hipStream_t)stream, points, (long)N, voxel_size, (long long*)bounds6_out, (unsigned int*)dist_max_out);
}
static int64_t voxel_slots(const int64_t* b) {
  int64_t v = 0;
  for (int a = 0; a < 3; ++a) v = (b[3 + a] - b[a]) > v ? (b[3 + a] - b[a]) : v;
  if (v > 2000000) return -1;
  const __int128 s = (__int128)v + (__int128)v * v + (__int128)v * v * v + 1;      // largest key gx + gy v + gz v^2 with g <= v, plus one
  return s > ((__int128)1 << 33) ? -1 : (int64_t)s;
}
extern "C" int64_t rap_voxel_table_slots(const int64_t* h_bounds6) { return h_bounds6 ? voxel_slots(h_bounds6) : -1; }
extern "C" size_t rap_voxel_workspace_bytes(const int64_t* h_bounds6) {
  const int64_t s = h_bounds6 ? voxel_slots(h_bounds6) : -1;
  if (s < 0) return 0;
  return align_up((size_t)s * 8, 256) + align_up((size_t)((s + 4095) / 4096) * 4, 256) + 256;
}
extern "C" int rap_voxel_downsample(const float* points, int64_t N, float voxel_size, const int64_t* h_bounds6, float dist_max,
                                    int64_t* indices_out, int32_t* count_out, void* ws, size_t ws_bytes, void* stream) {
  if (!points || !h_bounds6 || !indices_out || !count_out || N <= 0 || N > 0xffffffffLL || !(voxel_size > 0.f)) return RAP_ERR_INVALID;
  const int64_t slots = voxel_slots(h_bounds6);
  if (slots < 0) return RAP_ERR_INVALID;                 // grid too large for the dense table (> 2^33 slots = 64 GiB)
  if (!ws || ws_bytes < rap_voxel_workspace_bytes(h_bounds6)) return RAP_ERR_WORKSPACE;
  char* base = (char*)ws;
  unsigned long long* table = (unsigned long long*)base;
  unsigned int* block_cnt = (unsigned int*)(base + align_up((size_t)slots * 8, 256));
  return launch_voxel_downsample((hipStream_t)stream, points, (long)N, voxel_size, (const long long*)h_bounds6,
                                 dist_max > 0.f ? dist_max : 1.f, table, (long)slots, block_cnt, (unsigned int*)count_out,
                                 (long long*)indices_out);
}

// exact number of occupied voxels (calculate_voxel_coverage, point_sampling_utils.py:11-31): h_bounds6 from rap_voxel_bounds
static int64_t coverage_slots(const int64_t* b) {
  const __int128 s = (__int128)(b[3] - b[0] + 1) * (b[4] - b[1] + 1) * (b[5] - b[2] + 1);
  return (s <= 0 || s > ((__int128)1 << 36)) ? -1 : (int64_t)s;          // 64 GiB of one-byte slots
}
extern "C" size_t rap_voxel_coverage_workspace_bytes(const int64_t* h_bounds6) {
  const int64_t s = h_bounds6 ? coverage_slots(h_bounds6) : -1;
  return s < 0 ? 0 : align_up((size_t)s, 256) + 256;
}
extern "C" int rap_voxel_coverage(const float* points, int64_t N, float voxel_size, const int64_t* h_bounds6, int64_t* count_out, void* ws,
                                  size_t ws_bytes, void* stream) {
  if (!points || !h_bounds6 || !count_out || N <= 0 || !(voxel_size > 0.f)) return RAP_ERR_INVALID;
  const int64_t slots = coverage_slots(h_bounds6);
  if (slots < 0) return RAP_ERR_INVALID;
  if (!ws || ws_bytes < rap_voxel_coverage_workspace_bytes(h_bounds6)) return RAP_ERR_WORKSPACE;
  return launch_voxel_coverage((hipStream_t)stream, points, (long)N, voxel_size, (const long long*)h_bounds6, (unsigned char*)ws, (long)slots,
                               (unsigned long long*)count_out);
}

// O(N)-memory variants of the two calls above (voxel_sort.hip): identical results from a radix sort of per-point keys.  For grids whose
// dense table is large against N or beyond its 2^33 / 2^36-slot limits (a 100 m scene at 5 cm voxels = 64 GB of table).
extern "C" size_t rap_voxel_sorted_workspace_bytes(int64_t N) { return N > 0 && N <= 0xffffffffLL ? voxel_sorted_workspace_bytes((long)N) : 0; }
extern "C" int rap_voxel_downsample_sorted(const float* points, int64_t N, float voxel_size, const int64_t* h_bounds6, float dist_max,
                                           int64_t* indices_out, int32_t* count_out, void* ws, size_t ws_bytes, void* stream) {
  if (!points || !h_bounds6 || !indices_out || !count_out || N <= 0 || N > 0xffffffffLL || !(voxel_size > 0.f)) return RAP_ERR_INVALID;
  if (!ws) return RAP_ERR_WORKSPACE;
  return launch_voxel_downsample_sorted((hipStream_t)stream, points, (long)N, voxel_size, (const long long*)h_bounds6,
                                        dist_max > 0.f ? dist_max : 1.f, ws, ws_bytes, (long long*)indices_out, count_out);
}
extern "C" int rap_voxel_coverage_sorted(const float* points, int64_t N, float voxel_size, const int64_t* h_bounds6, int64_t* count_out,
                                         void* ws, size_t ws_bytes, void* stream) {
  if (!points || !h_bounds6 || !count_out || N <= 0 || N > 0xffffffffLL || !(voxel_size > 0.f)) return RAP_ERR_INVALID;
  if (!ws) return RAP_ERR_WORKSPACE;
  return launch_voxel_coverage_sorted((hipStream_t)stream, points, (long)N, voxel_size, (const long long*)h_bounds6, ws, ws_bytes,
                                      (long long*)count_out);
}

// ---------------------------------------------------------------------------------------------
// input side of the boundary: raw parts -> the packed batch
// ---------------------------------------------------------------------------------------------
extern "C" size_t rap_collate_workspace_bytes(int32_t B, int32_t P) {
  if (B <= 0 || P <= 0) return 0;
  return collate_workspace_bytes(B, P);
}

extern "C" int rap_collate_transform(const void* points, int32_t points_are_f64, const int64_t* points_per_part, int32_t B, int32_t P,
                                     int64_t TP, const int64_t* order, const float* feat_in, int32_t F, float* cond, float* gt,
                                     float* feat_out, uint8_t* anchor_indices, int64_t* part_indices, float* rotations,
                                     float* translations, float* scales, uint8_t* anchor_parts, float* global_translation,
                                     int64_t* cu_seqlens, int32_t* order_flag, void* ws, size_t ws_bytes, void* stream) {
  if (!points || !points_per_part || !cond || !gt || !anchor_indices || !part_indices || !rotations || !translations || !scales ||
      !anchor_parts || !global_translation || !cu_seqlens)
    return RAP_ERR_INVALID;
  if (B <= 0 || P <= 0 || P > 256 || (int64_t)B * P > 65535 || TP < 0 || TP > 0x7fffffffLL / 8 || F < 0) return RAP_ERR_INVALID;
  if (F > 0 && (!feat_in || !feat_out)) return RAP_ERR_INVALID;
  if (!ws || ws_bytes < collate_workspace_bytes(B, P)) return RAP_ERR_WORKSPACE;
  return launch_collate_transform((hipStream_t)stream, points, points_are_f64 ? 1 : 0, points_per_part, B, P, (long)TP, order, feat_in, F,
                                  cond, gt, feat_out, anchor_indices, part_indices, rotations, translations, scales, anchor_parts,
                                  global_translation, cu_seqlens, order_flag, ws);
}

// Consistency of a packed batch: see check_batch_kernel.  flag_out: device int32 (0 = consistent).
extern "C" int rap_check_batch(const int64_t* points_per_part, const int32_t* cu_batch, int32_t B, int32_t P, int64_t TP,
                               int32_t* flag_out, void* stream) {
  if (!points_per_part || !cu_batch || !flag_out || B <= 0 || P <= 0 || TP < 0) return RAP_ERR_INVALID;
  return launch_check_batch((hipStream_t)stream, points_per_part, cu_batch, B, P, (long)TP, flag_out);
}

extern "C" int rap_poison_on_flag(const int32_t* flag, float* buf, int64_t n, void* stream) {
  if (!flag || !buf || n < 0) return RAP_ERR_INVALID;
  return launch_poison_on_flag((hipStream_t)stream, flag, buf, (long)n);
}

// ---------------------------------------------------------------------------------------------
// statistical outlier removal (preprocessing)
// ---------------------------------------------------------------------------------------------
extern "C" size_t rap_outlier_workspace_bytes(int64_t N) { return N <= 0 ? 0 : outlier_workspace_bytes((long)N); }

extern "C" int rap_statistical_outliers(const float* points, int64_t N, int32_t nb_neighbors, double std_ratio, int64_t* inlier_indices,
                                        int32_t* count_out, double* stats_out, void* ws, size_t ws_bytes, void* stream) {
  if (!points || !inlier_indices || !count_out || N <= 0 || N > 0x7fffffffLL / 8 || nb_neighbors < 1 || nb_neighbors > 32 || !(std_ratio > 0.0))
    return RAP_ERR_INVALID;
  if (!ws || ws_bytes < outlier_workspace_bytes((long)N)) return RAP_ERR_WORKSPACE;
  return launch_statistical_outliers((hipStream_t)stream, points, (long)N, nb_neighbors, std_ratio, inlier_indices, count_out, stats_out, ws);
}

// QKV projection with MultiHeadRMSNorm fused into the epilogue (EPI_H_QKV_NORM): see rapflow.h
extern "C" int rap_gemm_h16_qkvnorm(int32_t dtype, const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, uint16_t* qk_out,
                                    int32_t M, int32_t K, int32_t heads, const float* gamma_q, const float* gamma_k, float q_mul,
                                    uint16_t* vt, int32_t vt_nblk, void* stream) {
  if (!A || !W || !qk_out || !gamma_q || !gamma_k || !vt || heads <= 0) return RAP_ERR_INVALID;
  GemmParamsH g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = qk_out; g.M = M; g.N = 3 * heads * 64; g.K = K; g.heads = heads;
  g.vt = vt; g.vt_nblk = vt_nblk; g.gamma_q = gamma_q; g.gamma_k = gamma_k; g.q_mul = q_mul;
  return launch_gemm_h16((hipStream_t)stream, dtype, EPI_H_QKV_NORM, g);
}
