"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's sampling hot path.

This file is the checker the HIP path is compared with; it is never imported by
``rap_amd/`` (the product) and is never the thing measured, except as the
``cpu_baseline`` leg of ``bench.py``.

It restates, function by function, the arithmetic of

* ``rectified_point_flow/sampler.py``            (flow_sampler, euler_step)
* ``rectified_point_flow/flow_model/*.py``       (PointCloudDiT and its layers)
* ``rectified_point_flow/procrustes.py``         (solve_procrustes, fit_transformations, rigidify)
* ``rectified_point_flow/utils/point_clouds.py`` (split_parts, repeat_by_cu_seqlens)
* the closure of ``rectified_point_flow/modeling.py:659-722`` (sample_rectified_flow)

as plain functional torch-CPU code over a ``state_dict`` (fp32 or fp64), each
function citing the reference lines it follows.  Two third-party pieces whose
source is not in the mount are restated from their published behaviour:
``flash_attn.flash_attn_varlen_qkvpacked_func`` (flash-attn 2.7.4.post1,
install.sh:19) and diffusers 0.33.0 ``FeedForward("geglu")`` / ``Timesteps`` /
``TimestepEmbedding`` (install.sh:9) -- those two are "parity unpinned" (the
reference ships no test for them).

Pinning: ``tests/test_oracle.py`` executes the reference's own
unmodified modules (``oracle/ref_loader.py``) on the same inputs and requires
agreement to fp32 round-off; ``tests/golden/*.npz`` were produced by the
reference's modules (``oracle/make_golden.py``) and are checked against this
file on every CPU test run (the mount does not exist on the GPU box).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# utils/point_clouds.py
# ----------------------------------------------------------------------------
def repeat_by_cu_seqlens(x: torch.Tensor, cu_seqlens: torch.Tensor) -> torch.Tensor:
    """point_clouds.py:161-184 -- row b of x repeated (cu[b+1]-cu[b]) times."""
    lens = (cu_seqlens[1:] - cu_seqlens[:-1]).to(device=x.device, dtype=torch.int64)
    idx = torch.repeat_interleave(torch.arange(x.shape[0], device=x.device), lens)
    return x.index_select(0, idx)


def split_parts(pointclouds: torch.Tensor, points_per_part: torch.Tensor, cu_seqlens_batch: torch.Tensor):
    """point_clouds.py:6-48 (packed branch): list over samples of list over non-empty parts."""
    if cu_seqlens_batch is None:
        raise ValueError("cu_seqlens_batch is required when pointclouds has shape (TP, 3)")
    out = []
    counts_per_batch = points_per_part.tolist()
    for b, counts in enumerate(counts_per_batch):
        seg = pointclouds[int(cu_seqlens_batch[b]):int(cu_seqlens_batch[b + 1])]
        assert sum(counts) == seg.size(0), "Mismatch detected: sum(counts) != segment length"
        out.append([s for s in torch.split(seg, counts, dim=0) if s.size(0) > 0])
    return out


# ----------------------------------------------------------------------------
# flow_model/embedding.py
# ----------------------------------------------------------------------------
def posenc(x: torch.Tensor, num_freqs: int = 10) -> torch.Tensor:
    """embedding.py:29-58: [x, sin(f0 x), cos(f0 x), ..., sin(f9 x), cos(f9 x)], f_k = 2^k."""
    outs = [x]
    freqs = 2.0 ** torch.linspace(0.0, num_freqs - 1, steps=num_freqs, device=x.device)
    for f in freqs:
        fx = x * f.to(x.dtype)
        outs.append(torch.sin(fx))
        outs.append(torch.cos(fx))
    return torch.cat(outs, -1)


def encoding_manager(sd, x, cond, feats, scales_pt, cfg=None, latent=None):
    """embedding.py:131-179: cat[PE63(cond), PE63(x), latent (TP, in_dim) if given, PE21(scale) if scale_emb_on, feat if
    local_feat_concat_on] -> emb_proj."""
    cfg = cfg or {}
    cols = [posenc(cond), posenc(x)]
    if latent is not None:                                                    # embedding.py:163-166
        cols.append(latent.view(x.shape[0], -1))
    if cfg.get("scale_emb_on", True):                                         # embedding.py:169-172
        cols.append(posenc(scales_pt.unsqueeze(-1)))
    if cfg.get("local_feat_concat_on", True) and feats is not None:           # embedding.py:175-177
        cols.append(feats)
    return F.linear(torch.cat(cols, dim=-1), sd["encoding_manager.emb_proj.weight"], sd["encoding_manager.emb_proj.bias"])


# ----------------------------------------------------------------------------
# flow_model/norm.py (+ diffusers Timesteps / TimestepEmbedding)
# ----------------------------------------------------------------------------
def timestep_sinusoid(t: torch.Tensor, num_channels: int = 256) -> torch.Tensor:
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0, scale=1)
    as configured at norm.py:50-52: [cos(t w_i), sin(t w_i)], w_i = exp(-ln(1e4) i/128).
    The frequencies are built in fp32 (as diffusers does) and the product/sin/cos in fp32."""
    half = num_channels // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32, device=t.device) / half
    w = torch.exp(exponent)
    arg = t[:, None].float() * w[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


def adaln_scale_shift(sd, prefix: str, t: torch.Tensor):
    """norm.py:71-73: emb = linear(SiLU(linear_2(SiLU(linear_1(TS(t)))))); scale, shift = chunk."""
    dt = sd[prefix + "linear.weight"].dtype
    ts = timestep_sinusoid(t).to(dt)
    h = F.linear(ts, sd[prefix + "timestep_embedder.linear_1.weight"], sd[prefix + "timestep_embedder.linear_1.bias"])
    h = F.linear(F.silu(h), sd[prefix + "timestep_embedder.linear_2.weight"], sd[prefix + "timestep_embedder.linear_2.bias"])
    e = F.linear(F.silu(h), sd[prefix + "linear.weight"], sd[prefix + "linear.bias"])
    return e.chunk(2, dim=-1)


def adaptive_layer_norm(sd, prefix, x, t, cu_batch):
    """norm.py:60-76: LN(no affine, eps 1e-5)(x) * (1 + scale_b) + shift_b."""
    scale, shift = adaln_scale_shift(sd, prefix, t)
    scale = repeat_by_cu_seqlens(scale, cu_batch)
    shift = repeat_by_cu_seqlens(shift, cu_batch)
    return F.layer_norm(x, (x.shape[-1],), eps=1e-5) * (1 + scale) + shift


def multi_head_rms_norm(x, gamma):
    """norm.py:28-33: F.normalize(x, dim=-1, eps=1e-12) * gamma * sqrt(Dh)."""
    return F.normalize(x, dim=-1) * gamma * (x.shape[-1] ** 0.5)


# ----------------------------------------------------------------------------
# flow_model/layer.py (+ flash-attn varlen, diffusers FeedForward)
# ----------------------------------------------------------------------------
def varlen_attention(qkv: torch.Tensor, cu_seqlens: torch.Tensor) -> torch.Tensor:
    """flash_attn_varlen_qkvpacked_func(qkv (T,3,H,D), cu_seqlens, softmax_scale=D^-1/2,
    causal=False, softcap=0, dropout=0) as called at layer.py:106-111,123-128."""
    T, _, H, D = qkv.shape
    out = torch.zeros((T, H, D), dtype=qkv.dtype, device=qkv.device)
    cu = cu_seqlens.tolist()
    for s in range(len(cu) - 1):
        a, b = cu[s], cu[s + 1]
        if b == a:
            continue
        q, k, v = (qkv[a:b, i].transpose(0, 1) for i in range(3))
        if qkv.is_cuda:
            out[a:b] = _device_attention(q, k, v).transpose(0, 1)
        else:
            out[a:b] = F.scaled_dot_product_attention(q, k, v).transpose(0, 1)
    return out


_DEVICE_SDPA = {"ok": None}


def _device_attention(q, k, v):
    """Attention of the DEVICE-side checker (tests only).  torch's memory-efficient SDPA kernel evaluates fp32 attention without
    materialising the score matrix -- measured on MI355X (profiles/r04_c2_device_checker_sdpa_probe.jsonl): 88 ms vs 199 ms for one
    8 x 65 536 x 64 problem, 3.0e-7 vs 2.8e-7 from fp64 -- which halves the time of the full-configuration parity tests.  Used when
    this torch build has it for the dtype (fp32; the flash backend is 16-bit only), else the explicit chunked form below; either way
    the device evaluation is pinned to the CPU one by tests/test_fullconfig_gpu.py::test_device_oracle_equals_cpu_oracle."""
    if _DEVICE_SDPA["ok"] is not False and q.dtype == torch.float32:
        try:
            from torch.nn.attention import SDPBackend, sdpa_kernel
            with sdpa_kernel([SDPBackend.EFFICIENT_ATTENTION]):
                o = F.scaled_dot_product_attention(q[None], k[None], v[None])[0]
            _DEVICE_SDPA["ok"] = True
            return o
        except (RuntimeError, ImportError):
            if _DEVICE_SDPA["ok"]:          # worked before: this shape is the problem (e.g. a length the kernel rejects)
                return _softmax_attention_chunked(q, k, v)
            _DEVICE_SDPA["ok"] = False
    return _softmax_attention_chunked(q, k, v)


def _softmax_attention_chunked(q, k, v, max_score_elems: int = 1 << 28):
    """The device-side checker's attention (tests only, see ``sample(device=...)``): softmax(q k^T / sqrt(D)) v written out with
    plain matmuls in the tensors' own dtype (fp32 / fp64), query rows in chunks so that the (H, chunk, L) score block stays below
    ``max_score_elems`` (1 GiB of fp32) -- the math the CPU branch's F.scaled_dot_product_attention evaluates, without depending on
    which fused SDPA backend a GPU build of torch would pick.  q, k, v: (H, L, D)."""
    H, L, D = q.shape
    scale = 1.0 / math.sqrt(D)
    # the softmax scale goes onto q before the product: for D = 64 it is 2^-3, an exact scaling of every product and partial sum,
    # so (q * scale) k^T == (q k^T) * scale bit for bit -- and the (H, chunk, L) score block is written once instead of twice
    qs = q * scale
    kt = k.transpose(1, 2)
    out = torch.empty_like(q)
    chunk = max(1, min(L, max_score_elems // max(1, H * L)))
    for a in range(0, L, chunk):
        p = torch.softmax(torch.matmul(qs[:, a:a + chunk], kt), dim=-1)
        out[:, a:a + chunk] = torch.matmul(p, v)
    return out


def attention_block(sd, prefix, which, x, cu_seqlens, H):
    """layer.py:98-131: qkv_proj (no bias) -> view (T,3,H,Dh) -> qk-norm -> varlen attn -> out_proj."""
    T, d = x.shape
    qkv = F.linear(x, sd[prefix + f"{which}_qkv_proj.weight"]).reshape(T, 3, H, d // H)
    q, k, v = qkv.unbind(dim=1)                                               # layer.py:91-96
    if prefix + f"{which}_q_norm.gamma" in sd:                                # layer.py:103-104: only with qk_norm=True
        q = multi_head_rms_norm(q, sd[prefix + f"{which}_q_norm.gamma"])
        k = multi_head_rms_norm(k, sd[prefix + f"{which}_k_norm.gamma"])
    out = varlen_attention(torch.stack([q, k, v], dim=1), cu_seqlens).reshape(T, d)
    return F.linear(out, sd[prefix + f"{which}_out_proj.weight"], sd[prefix + f"{which}_out_proj.bias"])


def feed_forward(sd, prefix, x):
    """diffusers FeedForward(dim, activation_fn="geglu") (layer.py:89,164):
    u = proj(x); h, g = chunk(u); y = h * gelu_erf(g); out = net.2(y)."""
    u = F.linear(x, sd[prefix + "ff.net.0.proj.weight"], sd[prefix + "ff.net.0.proj.bias"])
    h, g = u.chunk(2, dim=-1)
    return F.linear(h * F.gelu(g), sd[prefix + "ff.net.2.weight"], sd[prefix + "ff.net.2.bias"])


def dit_layer(sd, i, h, t, cu_batch, cu_part, H, taps=None):
    """layer.py:134-166."""
    p = f"transformer_layers.{i}."
    x = adaptive_layer_norm(sd, p + "self_prenorm.", h, t, cu_batch)
    if taps is not None and i == 0:
        taps["l0_self_prenorm"] = x.clone()
    a = attention_block(sd, p, "self", x, cu_part, H)
    h = h + a
    if taps is not None and i == 0:
        taps["l0_after_part_attn"] = h.clone()
    x = adaptive_layer_norm(sd, p + "global_prenorm.", h, t, cu_batch)
    h = h + attention_block(sd, p, "global", x, cu_batch, H)
    if taps is not None and i == 0:
        taps["l0_after_global_attn"] = h.clone()
    x = F.layer_norm(h, (h.shape[-1],), sd[p + "ff_norm.weight"], sd[p + "ff_norm.bias"], eps=1e-5)
    h = h + feed_forward(sd, p, x)
    if taps is not None and i == 0:
        taps["l0_out"] = h.clone()
    return h


# ----------------------------------------------------------------------------
# flow_model/point_cloud_dit.py
# ----------------------------------------------------------------------------
def dit_forward(sd, cfg, x, timesteps, cond, feats, scales, anchor, cu_batch, cu_part,
                return_transformer_features: bool = False, taps: dict | None = None, latent=None):
    """PointCloudDiT.forward (point_cloud_dit.py:141-191); ``latent`` = its latent_features argument (models with in_dim > 0)."""
    scales_pt = repeat_by_cu_seqlens(scales, cu_batch)                        # :174
    h = encoding_manager(sd, x, cond, feats, scales_pt, cfg, latent)          # :175
    emb = sd["anchor_part_emb.weight"]
    h = h + torch.where(anchor[:, None], emb[1][None, :], emb[0][None, :])    # :119-139
    if taps is not None:
        taps["embed"] = h.clone()
    for i in range(cfg["num_layers"]):                                        # :179-180
        h = dit_layer(sd, i, h, timesteps, cu_batch, cu_part, cfg["num_heads"], taps)
    y = F.silu(F.linear(h, sd["final_mlp.0.weight"], sd["final_mlp.0.bias"]))  # :111-117,183-184
    y = F.silu(F.linear(y, sd["final_mlp.2.weight"], sd["final_mlp.2.bias"]))
    v = F.linear(y, sd["final_mlp.4.weight"])
    if return_transformer_features:
        return {"velocity": v, "transformer_features": h}
    return v


# ----------------------------------------------------------------------------
# procrustes.py
# ----------------------------------------------------------------------------
def solve_procrustes(source: torch.Tensor, target: torch.Tensor):
    """procrustes.py:6-37.  Convention target ~= source @ R^T + t."""
    sm = source.mean(dim=0, keepdim=True)
    tm = target.mean(dim=0, keepdim=True)
    Hm = (source - sm).t() @ (target - tm)
    # device-side checker: the 3 x 3 factorisation runs on the host LAPACK exactly as in the CPU oracle (the moments above are the
    # device's work); a no-op for CPU tensors
    U, _, Vt = torch.linalg.svd(Hm.cpu())
    R = Vt.t() @ U.t()
    if torch.det(R) < 0:
        Vt = Vt.clone()
        Vt[-1, :] *= -1
        R = Vt.t() @ U.t()
    R = R.to(source.device)
    t = tm - sm @ R.t()
    return R, t.squeeze(0)


def fit_transformations(source, target, points_per_part, cu_seqlens_batch):
    """procrustes.py:40-84: zero rows for empty parts."""
    B, P = points_per_part.shape
    ps = split_parts(source, points_per_part, cu_seqlens_batch)
    pt = split_parts(target, points_per_part, cu_seqlens_batch)
    R = torch.zeros(B, P, 3, 3, dtype=source.dtype, device=source.device)
    t = torch.zeros(B, P, 3, dtype=source.dtype, device=source.device)
    points_per_part = points_per_part.cpu()
    for b in range(B):
        for p in range(P):
            if points_per_part[b, p] == 0:
                continue
            # NB (procrustes.py:79): the reference indexes the *compacted* per-sample list with p,
            # i.e. it assumes empty parts only ever trail the non-empty ones (the collate guarantees it).
            R[b, p], t[b, p] = solve_procrustes(ps[b][p], pt[b][p])
    return R, t


def rigidify_prediction_with_procrustes(prediction, condition, points_per_part, cu_seqlens_batch):
    """procrustes.py:86-118: out[off:off+n] = cond_p R^T + t in (b, p) order."""
    B, P = points_per_part.shape
    ps = split_parts(condition, points_per_part, cu_seqlens_batch)
    pt = split_parts(prediction, points_per_part, cu_seqlens_batch)
    out = torch.zeros_like(prediction)
    points_per_part = points_per_part.cpu()
    off = 0
    for b in range(B):
        for p in range(P):
            n = int(points_per_part[b, p])
            if n == 0:
                continue
            R, t = solve_procrustes(ps[b][p], pt[b][p])
            out[off:off + n] = ps[b][p] @ R.t() + t
            off += n
    return out


# ----------------------------------------------------------------------------
# sampler.py
# ----------------------------------------------------------------------------
def euler_step(x_t, t: float, dt: float, flow_model_fn):
    """sampler.py:79-92."""
    v = flow_model_fn(x_t, t)
    x0_hat = x_t - v * t
    x_t = x_t - dt * v
    return x_t, x0_hat


def flow_sampler(flow_model_fn, x_1, num_steps, points_per_part, cu_seqlens_batch, condition,
                 rigidity_forcing: bool):
    """sampler.py:11-74 with return_trajectory=True (the only way modeling.py:719 calls it)."""
    dt = 1.0 / num_steps
    x_t = x_1.clone()
    traj = torch.empty((num_steps, *x_1.shape), dtype=x_1.dtype, device=x_1.device)
    traj_xt = torch.empty((num_steps, *x_1.shape), dtype=x_1.dtype, device=x_1.device)
    for step in range(num_steps):
        t = 1 - step * dt
        x_t, x0_hat = euler_step(x_t, t, dt, flow_model_fn)
        if rigidity_forcing:
            x0_r = rigidify_prediction_with_procrustes(x0_hat, condition, points_per_part, cu_seqlens_batch)
            x_t = x0_r * (1 - t + dt) + x_1 * (t - dt)
        traj[step] = x0_hat
        traj_xt[step] = x_t
    return {"end_point_trajectory": traj, "trajectory": traj_xt}


# ----------------------------------------------------------------------------
# modeling.py:203-231, 632-741 (the closure around the sampler) + :389-391 (final poses)
# ----------------------------------------------------------------------------
def prepare_cu_seqlens(inputs):
    ppp = inputs["points_per_part"]
    cu_part = F.pad(torch.cumsum(ppp[ppp > 0], 0), (1, 0)).to(torch.int32)    # modeling.py:219-222
    cu_batch = inputs["cu_seqlens"].to(torch.int32)                            # modeling.py:223
    return cu_batch, cu_part


@torch.inference_mode()
def sample(sd, cfg, inputs, num_steps: int, rigidity_forcing: bool, dtype=torch.float32,
           max_steps: int | None = None, device=None, step_hook=None):
    """sample_rectified_flow + fit_transformations on the last end-point (modeling.py:356-391).

    ``max_steps`` (oracle-only convenience for the bounded CPU baseline): run only the first
    ``max_steps`` of ``num_steps`` flow steps (same dt and time grid).

    ``step_hook`` (bench.py's CPU leg): called with no arguments at the start of every flow step (= every model call).

    ``device`` (tests only): run this same restatement on a GPU through PyTorch-ROCm (plain fp32 / fp64 torch ops; attention by
    ``_softmax_attention_chunked``, the 3 x 3 SVDs on the host) as the DEVICE-SIDE CHECKER for the configurations the CPU cannot
    reach in test time (all 32 pairs of configs[1], all 50 steps of configs[4], 400 000-token samples).  It is pinned to the CPU
    evaluation of this file -- and through it to the reference -- by tests/test_fullconfig_gpu.py::test_device_oracle_equals_cpu_oracle."""
    device = torch.device("cpu") if device is None else torch.device(device)
    sd = {k: v.to(device=device, dtype=dtype) for k, v in sd.items()}
    cond = inputs["pointclouds"].to(device=device, dtype=dtype)
    feats = inputs["features"].to(device=device, dtype=dtype)
    scales = inputs["scales"].to(device=device, dtype=dtype)
    x_1 = inputs["x_1"].to(device=device, dtype=dtype)
    latent = inputs["latent_features"].to(device=device, dtype=dtype) if inputs.get("latent_features") is not None else None   # modeling.py:636
    anchor = inputs["anchor_indices"].to(device)
    ppp = inputs["points_per_part"].cpu()
    cu_batch, cu_part = prepare_cu_seqlens({"points_per_part": ppp, "cu_seqlens": inputs["cu_seqlens"].cpu()})
    B = cu_batch.shape[0] - 1

    # feature capture of the sampling call (modeling.py:666-708): the model call with index num_steps - 1 (or the first one with
    # t < 1e-6) runs with return_transformer_features=True; earlier calls count up
    captured = {"features": None}
    call_count = [0]

    def fn(x, t):
        if step_hook is not None:
            step_hook()
        ts = torch.full((B,), t, dtype=dtype, device=device)                   # modeling.py:674
        is_last_call = (t < 1e-6) or (call_count[0] >= num_steps - 1)          # modeling.py:678
        if is_last_call and captured["features"] is None:                      # modeling.py:680-695
            r = dit_forward(sd, cfg, x, ts, cond, feats, scales, anchor, cu_batch, cu_part, return_transformer_features=True, latent=latent)
            captured["features"] = r["transformer_features"]
            return r["velocity"]
        call_count[0] += 1                                                     # modeling.py:697
        return dit_forward(sd, cfg, x, ts, cond, feats, scales, anchor, cu_batch, cu_part, latent=latent)

    if max_steps is None:
        res = flow_sampler(fn, x_1, num_steps, ppp, cu_batch, cond, rigidity_forcing)
    else:
        dt = 1.0 / num_steps
        x_t = x_1.clone()
        traj = torch.empty((max_steps, *x_1.shape), dtype=dtype, device=device)
        traj_xt = torch.empty((max_steps, *x_1.shape), dtype=dtype, device=device)
        for step in range(max_steps):
            t = 1 - step * dt
            x_t, x0_hat = euler_step(x_t, t, dt, fn)
            if rigidity_forcing:
                x0_r = rigidify_prediction_with_procrustes(x0_hat, cond, ppp, cu_batch)
                x_t = x0_r * (1 - t + dt) + x_1 * (t - dt)
            traj[step] = x0_hat
            traj_xt[step] = x_t
        res = {"end_point_trajectory": traj, "trajectory": traj_xt}
    R, t = fit_transformations(cond, res["end_point_trajectory"][-1], ppp, cu_batch)
    return {"end_point_trajectory": res["end_point_trajectory"], "trajectory": res["trajectory"], "R": R, "t": t,
            "transformer_features": captured["features"]}     # None when max_steps stops before the capturing call


# ----------------------------------------------------------------------------
# eval/metrics.py:284-294 -- the SE(3) error formulae used to report pose deviation
# ----------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------------
# generation selection by rigidity
# ---------------------------------------------------------------------------------------------
def compute_rigidity_rmse(cond, pred, R, t, points_per_part, cu_seqlens_batch, scales=None, average_per_part=False):
    """eval/metrics.py:511-622: per object RMS of |cond_p R_p^T + t_p - pred_p| over all points of its non-empty parts
    (average_per_part: mean over parts of the per-part RMS); inf for an object without points; x scales."""
    B, P = points_per_part.shape
    parts_in = split_parts(cond, points_per_part, cu_seqlens_batch)
    parts_pr = split_parts(pred, points_per_part, cu_seqlens_batch)
    out = torch.zeros(B, dtype=cond.dtype)
    for b in range(B):
        sq, rm = [], []
        for p in range(P):
            if points_per_part[b, p] == 0:
                continue                                                          # metrics.py:566-567 / 594-595
            e = ((parts_in[b][p] @ R[b, p].T + t[b, p] - parts_pr[b][p]) ** 2).sum(dim=1)   # :577-581 / :605-609
            sq.append(e); rm.append(torch.sqrt(e.mean()))
        if not sq:
            out[b] = float("inf")                                                 # :589 / :616
        elif average_per_part:
            out[b] = torch.stack(rm).mean()                                       # :587
        else:
            out[b] = torch.sqrt(torch.cat(sq).mean())                             # :613-614
    return out * scales if scales is not None else out                            # :619-620


def average_trajectory_rigidity_rmse(cond, trajectory, points_per_part, cu_seqlens_batch, scales=None):
    """modeling.py:466-489: mean over trajectory steps of the rigidity RMSE of each step against its own Procrustes fit.
    -> (mean (B,), per_step (S,B))"""
    per_step = []
    for s in range(trajectory.shape[0]):
        R, t = fit_transformations(cond, trajectory[s], points_per_part, cu_seqlens_batch)          # :479-481
        per_step.append(compute_rigidity_rmse(cond, trajectory[s], R, t, points_per_part, cu_seqlens_batch, scales))
    per_step = torch.stack(per_step)
    return per_step.mean(dim=0), per_step                                                          # :489


def select_generations_by_rigidity(stacked_rigidity, final_clouds, rotations, translations, cu_seqlens_batch):
    """modeling.py:518, 560-592: per object the generation of smallest rigidity RMSE; gather its cloud / R / t."""
    best = torch.argmin(stacked_rigidity, dim=0)
    B = best.shape[0]
    cloud = torch.cat([final_clouds[int(best[b])][int(cu_seqlens_batch[b]):int(cu_seqlens_batch[b + 1])] for b in range(B)])
    R = torch.stack([rotations[int(best[b])][b] for b in range(B)])
    t = torch.stack([translations[int(best[b])][b] for b in range(B)])
    return best, cloud, R, t


def compute_overlap_ratio(pred, points_per_part, cu_seqlens_batch, taus=(0.005, 0.01, 0.02)):
    """eval/metrics.py:625-691 with exact fp64 pairwise distances (the reference's fp32 torch.cdist may use the
    |a|^2+|b|^2-2ab expansion, ~1e-7 absolute in d^2): -> (ratios (T,B), min_other_dist (TP,) fp64, inf where no other part)."""
    B, P = points_per_part.shape
    ratios = torch.zeros(len(taus), B, dtype=torch.float64)
    min_d = torch.full((pred.shape[0],), float("inf"), dtype=torch.float64)
    for b in range(B):
        a, e = int(cu_seqlens_batch[b]), int(cu_seqlens_batch[b + 1])
        if e <= a:
            continue
        pts = pred[a:e].double()
        pid = torch.repeat_interleave(torch.arange(P), points_per_part[b])                     # ppp_to_ids, point_clouds.py:86-91
        if e - a <= 1 or pid.unique().numel() <= 1:
            continue                                                                            # metrics.py:667-668
        d = torch.cdist(pts, pts, p=2, compute_mode="donot_use_mm_for_euclid_dist")
        d[pid[:, None] == pid[None, :]] = float("inf")                                          # :673-676
        m = d.min(dim=1).values
        min_d[a:e] = m
        for ti, tau in enumerate(taus):
            ratios[ti, b] = (m <= float(tau)).double().mean()                                   # :681-683
    return ratios, min_d


def compute_cd(gt, pred, cu_seqlens_batch):
    """eval/metrics.py:14-48.  pytorch3d 0.7.8 chamfer_distance(x, y, single_directional=False, norm=2, point_reduction="mean")
    is absent: its documented result is mean_i min_j |x_i - y_j|^2 + mean_j min_i |y_j - x_i|^2 (parity unpinned for that call);
    the reference then takes sqrt(0.5 * cd).  fp64 pairwise distances."""
    out = []
    for a, e in zip(cu_seqlens_batch[:-1].tolist(), cu_seqlens_batch[1:].tolist()):
        d = torch.cdist(gt[a:e].double(), pred[a:e].double(), p=2, compute_mode="donot_use_mm_for_euclid_dist") ** 2
        cd = d.min(dim=1).values.mean() + d.min(dim=0).values.mean()
        out.append((0.5 * cd).sqrt())
    return torch.stack(out)


def compute_correspondence_rmse(source_gt, target_gt, source_pred, target_pred, distance_threshold=0.1):
    """eval/metrics.py:386-469 with fp64 distances -> (rmse, num_correspondences, ratio, nearest indices)"""
    d = torch.cdist(source_gt.double(), target_gt.double(), p=2, compute_mode="donot_use_mm_for_euclid_dist")
    mind, nn = d.min(dim=1)
    valid = mind <= distance_threshold
    n = int(valid.sum())
    if n == 0:
        return torch.tensor(float("inf")), 0, 0.0, nn
    se = ((source_pred[valid].double() - target_pred[nn[valid]].double()) ** 2).sum(dim=1)
    return torch.sqrt(se.mean()), n, n / source_gt.shape[0], nn


def farthest_point_sampling(points, length, K, start):
    """pytorch3d 0.7.8 sample_farthest_points for ONE cloud (absent wheel; its published algorithm -- parity unpinned):
    selected[0] = start; repeat: closest_dists = min(closest_dists, |p - p_last|^2); next = argmax (first maximum).
    points (P,3) -> idx (min(K, length),) long.  fp32 distances like the kernel and pytorch3d."""
    p = points[:length].float()
    K = min(int(K), int(length))
    idx = torch.empty(K, dtype=torch.long)
    d = torch.full((length,), float("inf"))
    last = int(start)
    idx[0] = last
    for k in range(1, K):
        diff = p - p[last]
        d = torch.minimum(d, (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2])
        last = int(torch.argmax(d))
        idx[k] = last
    return idx


# ---------------------------------------------------------------------------------------------
# output transforms
# ---------------------------------------------------------------------------------------------
def relative_transforms(R_pred, t_pred, R_gt, t_gt, scales, points_per_part, global_rotation=None, global_translation=None):
    """eval/evaluator.py:436-474 for the whole batch -> (B,P,4,4) float32 (zero blocks for empty parts)."""
    import numpy as np
    B, P = points_per_part.shape
    out = np.zeros((B, P, 4, 4), dtype=np.float32)
    for b in range(B):
        s = float(scales[b])
        for p in range(P):
            if points_per_part[b, p] == 0:
                continue
            Rp, Rg = R_pred[b, p].numpy().astype(float), R_gt[b, p].numpy().astype(float)
            tp, tg = t_pred[b, p].numpy().astype(float) * s, t_gt[b, p].numpy().astype(float) * s        # :440-441
            RrT = Rg @ Rp.T                                                                               # :448
            M = np.eye(4, dtype=np.float32)
            M[:3, :3] = RrT.T; M[:3, 3] = tp - tg @ RrT                                                   # :449-455
            if global_rotation is not None and global_translation is not None:
                Gm = np.eye(4, dtype=np.float32)
                Gm[:3, :3] = global_rotation[b].numpy().astype(float); Gm[:3, 3] = global_translation[b].numpy().astype(float)
                M = M @ np.linalg.inv(Gm)                                                                 # :474
            out[b, p] = M
    return torch.from_numpy(out)


def rotation_error_deg(R_a: torch.Tensor, R_b: torch.Tensor) -> torch.Tensor:
    """acos((trace(R_a^T R_b) - 1) / 2) in degrees, clamped (metrics.py:289-291)."""
    tr = torch.einsum("...ij,...ij->...", R_a, R_b)
    return torch.rad2deg(torch.acos(((tr - 1) / 2).clamp(-1, 1)))


def compute_transform_errors(rotations_gt, translations_gt, rotations_pred, translations_pred, points_per_part, anchor_part,
                             matched_part_ids=None, scale=None):
    """eval/metrics.py:165-303 with use_icp=False: rotation error (degrees) and translation error of every non-anchor, non-empty part
    relative to the sample's FIRST anchor part (:239-257; identity when the sample has none, :258-263), means over those parts
    (:298-301; 0 / 0 = NaN).  -> (rot_mean (B,), trans_mean (B,), rot (B,P), trans (B,P))."""
    B, P = points_per_part.shape
    dt = rotations_gt.dtype
    if matched_part_ids is not None:                                            # :223-227: re-orders the PREDICTED poses
        bi = torch.arange(B)[:, None]
        rotations_pred, translations_pred = rotations_pred[bi, matched_part_ids], translations_pred[bi, matched_part_ids]
    scale = torch.ones(B, dtype=dt) if scale is None else scale.to(dt)
    rot = torch.zeros(B, P, dtype=dt); trans = torch.zeros(B, P, dtype=dt)
    for b in range(B):
        idx = anchor_part[b].nonzero().squeeze(1)
        if idx.numel() > 0:
            a = int(idx[0])
            Rg_inv, Rp_inv = rotations_gt[b, a].T, rotations_pred[b, a].T      # :250-256
            tg_inv, tp_inv = -Rg_inv @ translations_gt[b, a], -Rp_inv @ translations_pred[b, a]
        else:
            Rg_inv = Rp_inv = torch.eye(3, dtype=dt); tg_inv = tp_inv = torch.zeros(3, dtype=dt)
        for p in range(P):
            if points_per_part[b, p] == 0 or anchor_part[b, p]:                 # :266-267
                continue
            Rg_rel, tg_rel = Rg_inv @ rotations_gt[b, p], Rg_inv @ translations_gt[b, p] + tg_inv          # :282-286
            Rp_rel, tp_rel = Rp_inv @ rotations_pred[b, p], Rp_inv @ translations_pred[b, p] + tp_inv
            dR = Rg_rel.T @ Rp_rel                                              # :289
            rot[b, p] = torch.rad2deg(torch.acos(torch.clamp(0.5 * (torch.trace(dR) - 1), -1.0, 1.0)))     # :294-296
            trans[b, p] = torch.norm((tp_rel - tg_rel) * scale[b])              # :290, :299
    n = ((points_per_part != 0) & (~anchor_part)).sum(dim=1)
    return rot.sum(1) / n, trans.sum(1) / n, rot, trans


def voxel_down_sample(points, voxel_size):
    """dataset_process/utils/dataset_utils.py:279-322 (voxel_down_sample_torch) restated with explicit loops over voxels:
    per occupied voxel (key = gx + gy v + gz v^2, v = largest grid coordinate -- :301-303, wrap-around collisions included) the
    point with the smallest (quantised centre distance, index) pair (:296-299, :306-316), in ascending key order (:305).
    fp32 arithmetic in the reference's order."""
    import numpy as np
    p = np.asarray(points, dtype=np.float32)
    vs = np.float32(voxel_size)
    grid = np.floor(p / vs)                                              # :294
    center = (grid + np.float32(0.5)) * vs                               # :295
    d = p - center
    sq = d * d
    dist = np.sqrt((sq[:, 0] + sq[:, 1]) + sq[:, 2])                     # :296
    level = (dist / dist.max() * np.float32(999)).astype(np.int64)       # :297-299
    g = grid.astype(np.int64)
    g = g - g.min(axis=0)                                                # :293, :301 (floor is monotonic: floor(min) = min(floor))
    v = int(g.max())                                                     # :302
    key = g[:, 0] + g[:, 1] * v + g[:, 2] * v * v                        # :303
    best = {}
    for i in range(p.shape[0]):
        k = int(key[i]); cand = (int(level[i]), i)
        if k not in best or cand < best[k]:
            best[k] = cand
    return np.array([best[k][1] for k in sorted(best)], dtype=np.int64)


# ---------------------------------------------------------------------------------------------
# input side of the boundary: PointCloudDataset._transform in its evaluation form (no rotation /
# scale augmentation; reference data/dataset.py:733-900) + variable_collate_fn (data/datamodule.py:169-198).  float64 like the
# reference (numpy), cast to float32 at the end.
# ---------------------------------------------------------------------------------------------
def transform_sample(parts, features, max_parts, perms):
    """parts: list of (n_i,3) float64; features: list of (n_i,F); perms: list of within-part permutations (what
    np.random.permutation returned for every part, dataset.py:819).  Returns the reference's per-sample result dict (the
    arithmetic keys)."""
    import numpy as np
    counts = np.array([len(p) for p in parts])
    offsets = np.concatenate([[0], np.cumsum(counts)])
    pts_gt = np.concatenate(parts).astype(np.float64)
    feats = np.concatenate(features)
    n_parts = len(parts)
    tran_global = pts_gt.mean(axis=0)                                   # :759
    primary = int(np.argmax(counts))                                    # :764
    st, ed = offsets[primary], offsets[primary + 1]
    primary_trans = pts_gt[st:ed].mean(axis=0)                          # :769 center_pcd
    scale = np.max(np.abs(pts_gt[st:ed] - primary_trans)) * 1.5         # :783
    pts_gt = (pts_gt - primary_trans) / scale                           # :791-794 (rot_global = I)
    gt_trans = pts_gt.mean(axis=0)                                      # :796 center_pcd
    pts_gt = pts_gt - gt_trans
    pts = pts_gt.copy()
    part_indices = np.zeros(len(pts), dtype=np.int64)
    trans = np.zeros((max_parts, 3), dtype=np.float32); rots = np.zeros((max_parts, 3, 3), dtype=np.float32)
    feats_out = feats.copy()
    for i in range(n_parts):                                            # _proc_part :802-830
        a, b = offsets[i], offsets[i + 1]
        c = pts_gt[a:b].mean(axis=0)
        part = pts_gt[a:b] - c
        o = perms[i]
        pts[a:b] = part[o]
        pts_gt[a:b] = pts_gt[a:b][o]
        feats_out[a:b] = feats[a:b][o]
        part_indices[a:b] = i
        trans[i] = c; rots[i] = np.eye(3)
    anchor = np.zeros(max_parts, bool); anchor[primary] = True          # :841-842
    anchor_indices = np.zeros(len(pts), bool)
    anchor_indices[st:ed] = True                                        # :860-870
    rots[primary] = np.eye(3); trans[primary] = -gt_trans
    pts[st:ed] = pts_gt[st:ed] + gt_trans
    ppp = np.zeros(max_parts, dtype=np.int64); ppp[:n_parts] = counts
    return {"pointclouds": pts.astype(np.float32), "pointclouds_gt": pts_gt.astype(np.float32), "features": feats_out.astype(np.float32),
            "rotations": rots, "translations": trans, "points_per_part": ppp, "part_indices": part_indices,
            "scales": np.array(scale, dtype=np.float32), "anchor_parts": anchor, "anchor_indices": anchor_indices,
            "init_rotation": np.eye(3, dtype=np.float32), "global_rotation": np.eye(3, dtype=np.float32),
            "global_translation": tran_global.astype(np.float32)}


def transform_and_collate(samples, max_parts, seed):
    """samples: list of {"parts": [...], "features": [...]}; draws the permutations like the reference (np.random.seed(seed), one
    np.random.permutation per part in order) and collates like variable_collate_fn."""
    import numpy as np
    np.random.seed(seed)
    outs = []
    for smp in samples:
        perms = [np.random.permutation(len(p)) for p in smp["parts"]]
        outs.append(transform_sample(smp["parts"], smp["features"], max_parts, perms))
    lengths = [o["pointclouds"].shape[0] for o in outs]
    res = {"cu_seqlens": torch.cat([torch.zeros(1, dtype=torch.int64), torch.tensor(lengths, dtype=torch.int64).cumsum(0)])}
    for k in ("pointclouds", "pointclouds_gt", "part_indices", "anchor_indices", "features"):
        res[k] = torch.from_numpy(np.concatenate([o[k] for o in outs], axis=0))
    for k in ("rotations", "translations", "points_per_part", "anchor_parts", "init_rotation", "scales", "global_rotation", "global_translation"):
        res[k] = torch.stack([torch.from_numpy(np.asarray(o[k])) for o in outs], dim=0)
    return res


# ---------------------------------------------------------------------------------------------
# preprocessing in front of FPS / MiniSpinNet
# ---------------------------------------------------------------------------------------------
def remove_statistical_outlier(points, nb_neighbors=20, std_ratio=2.5):
    """Open3D 0.18 PointCloud::RemoveStatisticalOutliers restated from its published source (the wheel is absent: PARITY UNPINNED;
    call site dataset_process/extract_sample_features.py:378-385): KNN of every point in its own cloud (the point itself is the
    first hit), mean of the sqrt distances; cloud mean over the positive ones divided by N; Bessel-corrected std; keep
    0 < d < mean + std_ratio * std.  Returns (inlier indices ascending, mean distances float64)."""
    import numpy as np
    from scipy.spatial import cKDTree
    p = np.asarray(points, dtype=np.float64)
    n = len(p)
    k = min(nb_neighbors, n)
    d, _ = cKDTree(p).query(p, k=k)
    d = d.reshape(n, k)
    avg = d.sum(axis=1) / k
    pos = avg > 0
    mean = avg[pos].sum() / n
    std = np.sqrt(((avg[pos] - mean) ** 2).sum() / (n - 1)) if n > 1 else 0.0
    thr = mean + std_ratio * std
    return np.nonzero(pos & (avg < thr))[0], avg


def calculate_voxel_coverage(points, voxel_size):
    """dataset_process/utils/point_sampling_utils.py:11-31 (numpy, float64 division like the reference)."""
    import numpy as np
    if len(points) == 0:
        return 0
    return len(np.unique(np.floor(np.asarray(points) / voxel_size).astype(int), axis=0))


def calculate_adaptive_sample_count_per_part(parts_points, voxel_size, voxel_ratio, min_points_per_part, max_sample_points):
    """point_sampling_utils.py:33-84."""
    out = []
    for pts in parts_points:
        if len(pts) == 0:
            out.append(0)
            continue
        c = int(calculate_voxel_coverage(pts, voxel_size) * voxel_ratio)
        out.append(min(max_sample_points, min(len(pts), max(min_points_per_part, c))))
    return out
