"""TEST INFRASTRUCTURE ONLY -- import the *unmodified* reference modules on CPU.

Only usable inside the build container where ``/root/reference`` is mounted
(the GPU box has no such path; nothing in ``-m gpu`` tests, ``smoke()`` or
``bench.py`` calls this).  It is used by ``oracle/make_golden.py`` to produce
the fixtures under ``tests/golden/`` and by ``tests/test_oracle.py``
to pin ``oracle/rap_oracle.py`` (the restatement that travels) against the
reference's own source.

Recipe: the reference package cannot be imported
naively (``rectified_point_flow/__init__.py`` pulls lightning / hydra /
pytorch3d).  We pre-register *bare* package objects for
``rectified_point_flow`` and ``rectified_point_flow.utils`` whose ``__path__``
points into the mount, so that the sub-modules we need
(``sampler``, ``procrustes``, ``flow_model.*``, ``utils.point_clouds``) are
executed from the reference's own files without running either ``__init__``.

Two third-party wheels that are on the path but absent here are stubbed by
restating their published semantics (pins from reference
``scripts/install.sh:9,19``):

* ``flash_attn`` 2.7.4.post1 -- ``flash_attn_varlen_qkvpacked_func``:
  per-segment, non-causal softmax(q k^T / sqrt(D)) v.
* ``diffusers`` 0.33.0 -- ``FeedForward(activation_fn="geglu")``,
  ``Timesteps``, ``TimestepEmbedding``.

Those two stubs are "parity unpinned": the reference ships no test that pins
them; they follow the libraries' documented behaviour.
"""
from __future__ import annotations

import importlib
import math
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = os.environ.get("RAP_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "rectified_point_flow", "flow_model"))


# ----------------------------------------------------------------------------
# stubs
# ----------------------------------------------------------------------------
_FLASH_STUB_SCORE_BYTES = 1 << 31      # largest (H, rows, L) score block the flash-attn stand-in materialises at once
def _flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0,
                                      softmax_scale=None, causal=False,
                                      window_size=(-1, -1), softcap=0.0,
                                      alibi_slopes=None, deterministic=False,
                                      return_attn_probs=False):
    """qkv (T,3,H,D); cu_seqlens (S+1,) int32 -> (T,H,D).  Dense softmax attention
    inside every segment, none across segments."""
    assert dropout_p == 0.0 and not causal and softcap == 0.0
    T, three, H, D = qkv.shape
    assert three == 3
    scale = softmax_scale if softmax_scale is not None else D ** -0.5
    out = torch.empty((T, H, D), dtype=qkv.dtype, device=qkv.device)
    cu = cu_seqlens.tolist()
    for s in range(len(cu) - 1):
        a, b = cu[s], cu[s + 1]
        if b == a:
            continue
        q = qkv[a:b, 0].transpose(0, 1)  # (H,L,D)
        k = qkv[a:b, 1].transpose(0, 1)
        v = qkv[a:b, 2].transpose(0, 1)
        L = b - a
        # Rows of a softmax are independent, so long segments are evaluated in query chunks (same formula per row; keeps the
        # (H,L,L) score matrix of the BASELINE configs[3]/[4] geometries -- up to 137 GB -- out of memory).
        rows = L if H * L * L * qkv.element_size() <= _FLASH_STUB_SCORE_BYTES else max(1, _FLASH_STUB_SCORE_BYTES // (H * L * qkv.element_size()))
        for r0 in range(0, L, rows):
            att = torch.softmax((q[:, r0:r0 + rows] @ k.transpose(1, 2)) * scale, dim=-1)
            out[a + r0:min(b, a + r0 + rows)] = (att @ v).transpose(0, 1)
    return out


class _GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out, bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2, bias=bias)

    def forward(self, x):
        h = self.proj(x)
        h, gate = h.chunk(2, dim=-1)
        return h * F.gelu(gate)  # exact erf GELU


class _FeedForward(nn.Module):
    """diffusers.models.attention.FeedForward (mult=4), keys net.0.proj.*, net.2.*"""

    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu",
                 final_dropout=False, inner_dim=None, bias=True):
        super().__init__()
        assert activation_fn == "geglu"
        inner_dim = int(dim * mult) if inner_dim is None else inner_dim
        dim_out = dim if dim_out is None else dim_out
        self.net = nn.ModuleList([_GEGLU(dim, inner_dim, bias=bias), nn.Dropout(dropout),
                                  nn.Linear(inner_dim, dim_out, bias=bias)])

    def forward(self, x, *a, **k):
        for m in self.net:
            x = m(x)
        return x


class _Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift, scale=1):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift
        self.scale = scale

    def forward(self, timesteps):
        half = self.num_channels // 2
        exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
        exponent = exponent / (half - self.downscale_freq_shift)
        emb = torch.exp(exponent)
        emb = timesteps[:, None].float() * emb[None, :]
        emb = self.scale * emb
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip_sin_to_cos:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb


class _TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu"):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample, condition=None):
        # diffusers keeps the sinusoid in fp32 and lets the Linear's dtype decide;
        # cast so that an fp64 oracle run stays fp64 end to end.
        sample = sample.to(self.linear_1.weight.dtype)
        return self.linear_2(self.act(self.linear_1(sample)))


def _install_stubs():
    if "flash_attn" not in sys.modules:
        m = types.ModuleType("flash_attn")
        m.flash_attn_varlen_qkvpacked_func = _flash_attn_varlen_qkvpacked_func
        sys.modules["flash_attn"] = m
    if "diffusers" not in sys.modules:
        d = types.ModuleType("diffusers"); d.__path__ = []
        dm = types.ModuleType("diffusers.models"); dm.__path__ = []
        da = types.ModuleType("diffusers.models.attention"); da.FeedForward = _FeedForward
        de = types.ModuleType("diffusers.models.embeddings")
        de.Timesteps = _Timesteps; de.TimestepEmbedding = _TimestepEmbedding
        sys.modules.update({"diffusers": d, "diffusers.models": dm,
                            "diffusers.models.attention": da, "diffusers.models.embeddings": de})


def _install_pytorch3d_import_stub():
    """eval/metrics.py imports pytorch3d (0.7.8, install.sh:16) at module level for chamfer / ICP metrics that are NOT on
    the path; compute_rigidity_rmse (metrics.py:511-622) does not touch it.  The stub only lets the unmodified module
    import; calling the stubbed functions raises."""
    if "pytorch3d" in sys.modules:
        return
    def _absent(*a, **k):
        raise NotImplementedError("pytorch3d is not installed in this container (stub for import only)")
    p3 = types.ModuleType("pytorch3d"); p3.__path__ = []
    loss = types.ModuleType("pytorch3d.loss"); loss.__path__ = []
    ch = types.ModuleType("pytorch3d.loss.chamfer"); ch.chamfer_distance = _absent
    ops = types.ModuleType("pytorch3d.ops"); ops.iterative_closest_point = _absent
    sys.modules.update({"pytorch3d": p3, "pytorch3d.loss": loss, "pytorch3d.loss.chamfer": ch, "pytorch3d.ops": ops})


class _PermissiveMeta(type):
    """Class attributes of an import-only stand-in are again stand-ins (`trimesh.util.log.setLevel(...)` at import time)."""
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _permissive_class(name)


def _permissive_class(name):
    return _PermissiveMeta(name, (), {"__init__": lambda self, *a, **k: None})


class _PermissiveModule(types.ModuleType):
    """Import-only stand-in: any attribute is an empty class (for `from x import A, B, C` of render / Lightning helpers that
    are never called on the path)."""
    __path__: list = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _permissive_class(name)


def load_reference_evaluator():
    """The reference's unmodified eval/evaluator.py (for Evaluator._save_transformation_files, evaluator.py:383-490).  It
    imports lightning (2.5.2) and, through utils/render.py, pytorch3d.structures / .renderer -- none of which the transform
    writer touches; they are replaced by import-only stand-ins."""
    ref = load_reference()
    if getattr(ref, "evaluator", None) is None:
        for n in ("pytorch3d.structures", "pytorch3d.renderer", "pytorch3d.renderer.points", "pytorch3d.renderer.cameras", "lightning"):
            sys.modules.setdefault(n, _PermissiveModule(n))
        ref.evaluator = importlib.import_module("rectified_point_flow.eval.evaluator")
    return ref.evaluator


def reference_transformation_files(data, sample_dir, dataset_name, sample_indices, generation_idx, rotations_pred, translations_pred,
                                   global_rotation=None, global_translation=None):
    """Run the reference's own Evaluator._save_transformation_files for every sample of the batch (as Evaluator.run does,
    evaluator.py:370-381); returns {file name: 4x4 float64 array parsed back from the text it wrote}."""
    import numpy as np
    from pathlib import Path
    ev = load_reference_evaluator()
    sample_dir = Path(sample_dir); sample_dir.mkdir(parents=True, exist_ok=True)
    fake_self = types.SimpleNamespace()
    B = data["points_per_part"].shape[0]
    for idx in range(B):
        gr = None if global_rotation is None else global_rotation[idx]
        gt = None if global_translation is None else global_translation[idx]
        ev.Evaluator._save_transformation_files(fake_self, data, idx, sample_dir, dataset_name, int(sample_indices[idx]), generation_idx,
                                                rotations_pred, translations_pred, gr, gt)
    return {p.name: np.loadtxt(p) for p in sorted(sample_dir.glob("*_transform.txt"))}


def reference_spinnet_forward(sd, pts, kpts, des_r, perm_seed, is_aligned_to_global_z=True):
    """Run the reference's UNMODIFIED MiniSpinNet (dataset_process/utils/spinnet/*) on CPU.  Test-only shims: pytorch3d.ops.
    ball_query is the restatement of oracle/spinnet_oracle.py (the wheel is absent); `Tensor.cuda` is a no-op for the duration
    of the call (SPT hard-codes `.cuda()`, patch_embedder.py:176); numpy's global RNG is seeded so that the shuffle of
    select_patches (:99) is reproducible -- the permutation it draws is returned."""
    import numpy as np
    from oracle import spinnet_oracle as SO
    _install_pytorch3d_import_stub()

    def ball_query(p1, p2, K, radius, return_nn=True, **kw):
        idx, nn = zip(*[SO.ball_query_first_k(p1[b], p2[b], K, radius) for b in range(p1.shape[0])])
        idx = torch.stack(idx); nn = torch.stack(nn)
        d2 = ((p1[:, :, None, :] - nn) ** 2).sum(-1) * (idx >= 0)
        return d2, idx, nn
    sys.modules["pytorch3d.ops"].ball_query = ball_query
    for name, path in (("dataset_process", "dataset_process"), ("dataset_process.utils", "dataset_process/utils"),
                       ("dataset_process.utils.spinnet", "dataset_process/utils/spinnet"),
                       ("dataset_process.utils.spinnet.utils", "dataset_process/utils/spinnet/utils")):
        if name not in sys.modules:
            m = types.ModuleType(name); m.__path__ = [os.path.join(REFERENCE_ROOT, path)]
            sys.modules[name] = m
    pe = importlib.import_module("dataset_process.utils.spinnet.patch_embedder")
    common = importlib.import_module("dataset_process.utils.spinnet.utils.common")
    pe.ball_query = ball_query; common.ball_query = ball_query          # both modules did `from pytorch3d.ops import ball_query`
    net = pe.MiniSpinNet(des_r=des_r)
    net.load_state_dict(sd, strict=False)                               # num_batches_tracked buffers are not in the blob
    net.eval()
    np.random.seed(perm_seed)
    perm = np.random.choice(pts.shape[0], pts.shape[0], replace=False)
    np.random.seed(perm_seed)
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with torch.no_grad():
            out = net(pts[None], kpts[None], des_r, is_aligned_to_global_z)
    finally:
        torch.Tensor.cuda = orig_cuda
    return {"desc": out["desc"], "patches": out["patches"], "perm": perm, "R": out["R"]}


def load_reference_dataset_utils():
    """The reference's UNMODIFIED dataset_process/utils/dataset_utils.py (voxel_down_sample_torch, :279-322).  Test-only shim:
    `open3d` and `h5py` (absent wheels) are import-only stubs -- the functions used from this module are pure torch."""
    if not reference_available():
        raise RuntimeError(f"reference not mounted at {REFERENCE_ROOT}")
    for absent in ("open3d", "h5py"):                     # io_utils imports h5py at module level
        if absent not in sys.modules:
            sys.modules[absent] = types.ModuleType(absent)
    for name, path in (("dataset_process", "dataset_process"), ("dataset_process.utils", "dataset_process/utils")):
        if name not in sys.modules:
            m = types.ModuleType(name); m.__path__ = [os.path.join(REFERENCE_ROOT, path)]
            sys.modules[name] = m
    return importlib.import_module("dataset_process.utils.dataset_utils")


def load_reference_data():
    """The reference's UNMODIFIED data/dataset.py and data/datamodule.py (PointCloudDataset._transform, dataset.py:733-900;
    variable_collate_fn, datamodule.py:169-198).  Import-only stand-ins for h5py, trimesh (mesh IO: not touched by _transform) and
    lightning (the DataModule base class)."""
    if not reference_available():
        raise RuntimeError(f"reference not mounted at {REFERENCE_ROOT}")
    for absent in ("h5py", "trimesh", "trimesh.exchange", "trimesh.exchange.ply", "lightning"):
        sys.modules.setdefault(absent, _PermissiveModule(absent))
    pkg_dir = os.path.join(REFERENCE_ROOT, "rectified_point_flow")
    if "rectified_point_flow" not in sys.modules:
        pkg = types.ModuleType("rectified_point_flow"); pkg.__path__ = [pkg_dir]
        sys.modules["rectified_point_flow"] = pkg
    if "rectified_point_flow.data" not in sys.modules:
        dpk = types.ModuleType("rectified_point_flow.data"); dpk.__path__ = [os.path.join(pkg_dir, "data")]
        sys.modules["rectified_point_flow.data"] = dpk
    ns = types.SimpleNamespace()
    ns.dataset = importlib.import_module("rectified_point_flow.data.dataset")
    ns.datamodule = importlib.import_module("rectified_point_flow.data.datamodule")
    return ns


class _SequentialPool:
    """Stand-in for the dataset's ThreadPoolExecutor: parts are processed in order, so the np.random.permutation draws of
    _proc_part (dataset.py:819) come in part order and are reproducible from the seed."""
    def map(self, fn, it):
        return [fn(x) for x in it]

    def shutdown(self):
        pass


def reference_transform_and_collate(samples, max_parts, seed, dataset_name="synthetic"):
    """Run the reference's own PointCloudDataset._transform (evaluation split: no augmentation) on every sample and its own
    variable_collate_fn on the results.  samples: list of {"parts": [ (n_i,3) float64 arrays ], "features": [ (n_i,F) float32 ]}.
    numpy's global RNG is seeded once; the permutations the reference draws (one per part, in order) are reproduced by
    `np.random.seed(seed)` + the same sequence of np.random.permutation calls."""
    import numpy as np
    ns = load_reference_data()
    fake = types.SimpleNamespace(split="val", yaw_augmentation=False, roll_pitch_range=None, random_scale_range=None,
                                 pool=_SequentialPool(), max_parts=max_parts, multi_anchor=False, multi_anchor_random_rate=0.0,
                                 dataset_name=dataset_name, load_features=True)
    np.random.seed(seed)
    outs = []
    for idx, smp in enumerate(samples):
        data = {"index": idx, "name": f"sample_{idx}", "num_parts": len(smp["parts"]), "overlap_threshold": 0.05,
                "pointclouds_gt": [np.array(a, dtype=np.float64) for a in smp["parts"]],
                "pointclouds_normals_gt": [np.zeros((len(a), 3)) for a in smp["parts"]],
                "features": [np.array(f, dtype=np.float32) for f in smp["features"]], "is_pre_sampled": True}
        outs.append(ns.dataset.PointCloudDataset._transform(fake, data))
    return ns.datamodule.variable_collate_fn(outs)


_LOADED = None


def load_reference():
    """Returns a namespace with the reference's own (unmodified) objects."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not reference_available():
        raise RuntimeError(f"reference not mounted at {REFERENCE_ROOT}")
    _install_stubs()
    pkg_dir = os.path.join(REFERENCE_ROOT, "rectified_point_flow")
    pkg = types.ModuleType("rectified_point_flow"); pkg.__path__ = [pkg_dir]
    utils = types.ModuleType("rectified_point_flow.utils"); utils.__path__ = [os.path.join(pkg_dir, "utils")]
    sys.modules.setdefault("rectified_point_flow", pkg)
    sys.modules.setdefault("rectified_point_flow.utils", utils)
    ns = types.SimpleNamespace()
    ns.point_clouds = importlib.import_module("rectified_point_flow.utils.point_clouds")
    ns.procrustes = importlib.import_module("rectified_point_flow.procrustes")
    ns.sampler = importlib.import_module("rectified_point_flow.sampler")
    ns.flow_model = importlib.import_module("rectified_point_flow.flow_model")
    ns.PointCloudDiT = ns.flow_model.PointCloudDiT
    ns.get_sampler = ns.sampler.get_sampler
    ns.fit_transformations = ns.procrustes.fit_transformations
    ns.rigidify_prediction_with_procrustes = ns.procrustes.rigidify_prediction_with_procrustes
    # generation selection: the reference's own compute_rigidity_rmse
    _install_pytorch3d_import_stub()
    evalp = types.ModuleType("rectified_point_flow.eval"); evalp.__path__ = [os.path.join(pkg_dir, "eval")]
    sys.modules.setdefault("rectified_point_flow.eval", evalp)
    ns.metrics = importlib.import_module("rectified_point_flow.eval.metrics")
    ns.compute_rigidity_rmse = ns.metrics.compute_rigidity_rmse
    ns.compute_overlap_ratio = ns.metrics.compute_overlap_ratio
    ns.compute_correspondence_rmse = ns.metrics.compute_correspondence_rmse
    _LOADED = ns
    return ns


def reference_generation_selection(cond, trajectories, points_per_part, cu_seqlens_batch, scales, use_average=True):
    """The rigidity-selection block of the reference's test_step (modeling.py:456-592) driven with the reference's OWN
    fit_transformations and compute_rigidity_rmse; the surrounding LightningModule bookkeeping (logging, evaluator) is not
    importable here (lightning / hydra), so the ~25 lines of control flow are restated verbatim below.
    trajectories: list over generations of (S,TP,3) end-point trajectories."""
    ref = load_reference()
    n_rot, n_trans, rig = [], [], []
    with torch.inference_mode():
        for trajs in trajectories:
            R, t = ref.fit_transformations(cond, trajs[-1], points_per_part, cu_seqlens_batch)       # modeling.py:389-391
            n_rot.append(R); n_trans.append(t)
            if use_average:                                                                          # :466-500
                step_rmses = []
                for step_idx in range(trajs.shape[0]):
                    Rs, ts = ref.fit_transformations(cond, trajs[step_idx], points_per_part, cu_seqlens_batch)
                    step_rmses.append(ref.compute_rigidity_rmse(cond, trajs[step_idx], Rs, ts, points_per_part,
                                                                cu_seqlens_batch, scales))
                rig.append(torch.stack(step_rmses).mean(dim=0))
            else:                                                                                    # :501-504
                rig.append(ref.compute_rigidity_rmse(cond, trajs[-1], R, t, points_per_part, cu_seqlens_batch, scales))
        stacked = torch.stack(rig)                                                                   # (G,B)
        best = torch.argmin(stacked, dim=0)                                                          # :518
        B = best.shape[0]
        sel = [trajectories[int(best[b])][-1][cu_seqlens_batch[b]:cu_seqlens_batch[b + 1]] for b in range(B)]   # :560-573
        R_sel = torch.zeros_like(n_rot[0]); t_sel = torch.zeros_like(n_trans[0])                                # :583-588
        for b in range(B):
            R_sel[b] = n_rot[int(best[b])][b]; t_sel[b] = n_trans[int(best[b])][b]
    return {"rigidity_rmse": stacked, "best_gen_indices": best, "pointclouds_selected": torch.cat(sel, dim=0),
            "rotations_selected": R_sel, "translations_selected": t_sel, "n_rotations": torch.stack(n_rot),
            "n_translations": torch.stack(n_trans)}


def build_reference_dit(cfg, state_dict, dtype=torch.float32):
    """Instantiate the reference's PointCloudDiT and load ``state_dict`` into it."""
    ns = load_reference()
    m = ns.PointCloudDiT(in_dim=int(cfg.get("in_dim", 0)), out_dim=3, embed_dim=cfg["embed_dim"], num_layers=cfg["num_layers"],
                         num_heads=cfg["num_heads"], attn_dtype="float32", qk_norm=cfg.get("qk_norm", True),
                         local_feat_dim=cfg["local_feat_dim"], scale_emb_on=cfg.get("scale_emb_on", True),
                         local_feat_concat_on=cfg.get("local_feat_concat_on", True))
    # fp32 only: the reference forces its head to fp32 (point_cloud_dit.py:183-184 `embed.float()`), so an
    # fp64 ground truth cannot be produced by the unmodified modules; oracle/rap_oracle.py (pinned to this
    # loader in fp32) provides the fp64 ground truth instead.
    assert dtype == torch.float32
    missing, unexpected = m.load_state_dict({k: v.to(dtype) for k, v in state_dict.items()}, strict=True)
    return m.eval()


def reference_sample(cfg, state_dict, inputs, num_steps, rigidity_forcing, dtype=torch.float32):
    """Restates only the 30-line closure of modeling.py:659-722 around the reference's own
    get_sampler / PointCloudDiT / procrustes (modeling.py itself needs lightning), INCLUDING its feature capture
    (modeling.py:666-708): the call with index num_steps - 1 (or the first one with t < 1e-6) runs the model with
    return_transformer_features=True and keeps `transformer_features`; earlier calls count up."""
    ns = load_reference()
    model = build_reference_dit(cfg, state_dict, dtype)
    cond = inputs["pointclouds"].to(dtype)
    feats = inputs["features"].to(dtype)
    scales = inputs["scales"].to(dtype)
    anchor = inputs["anchor_indices"]
    ppp = inputs["points_per_part"]
    x_1 = inputs["x_1"].to(dtype)
    latent = inputs["latent_features"].to(dtype) if inputs.get("latent_features") is not None else None     # modeling.py:636 (in_dim > 0)
    valid = ppp > 0
    cu_part = F.pad(torch.cumsum(ppp[valid], 0), (1, 0)).to(torch.int32)   # modeling.py:219-222
    cu_batch = inputs["cu_seqlens"].to(torch.int32)                          # modeling.py:223
    B = cu_batch.shape[0] - 1
    captured = {"features": None, "t": None}
    call_count = [0]                                                         # modeling.py:668

    @torch.inference_mode()
    def run():
        def fn(x, t):                                                        # modeling.py:672-708
            ts = torch.full((B,), t, dtype=dtype)
            kw = dict(x=x, timesteps=ts, cond_coord=cond, local_features=feats, latent_features=latent, scales=scales,
                      anchor_indices=anchor, cu_seqlens_batch=cu_batch, cu_seqlens_part=cu_part)
            is_last_call = (t < 1e-6) or (call_count[0] >= num_steps - 1)    # modeling.py:678
            if is_last_call and captured["features"] is None:                # modeling.py:680-695
                result = model(return_transformer_features=True, **kw)
                captured["features"] = result["transformer_features"]
                captured["t"] = float(t)
                return result["velocity"]
            call_count[0] += 1                                               # modeling.py:697
            return model(**kw)
        res = ns.get_sampler("euler")(flow_model_fn=fn, x_1=x_1, x_0=cond, condition=cond, points_per_part=ppp,
                                      cu_seqlens_batch=cu_batch, anchor_indices=anchor, num_steps=num_steps,
                                      return_trajectory=True, rigidity_forcing=rigidity_forcing)
        R, t = ns.fit_transformations(cond, res["end_point_trajectory"][-1], ppp, cu_batch)  # modeling.py:389-391
        return res, R, t
    res, R, t = run()
    return {"end_point_trajectory": res["end_point_trajectory"], "trajectory": res["trajectory"], "R": R, "t": t,
            "transformer_features": captured["features"], "features_timestep": captured["t"]}


class _StopAfter(Exception):
    pass


def reference_time_steps(cfg, state_dict, inputs, num_steps, rigidity_forcing, max_steps=None, dtype=torch.float32):
    """bench.py's CPU-baseline leg when the mount is present: the reference's UNMODIFIED sampler loop (sampler.py:11-74) around its
    own PointCloudDiT and procrustes, with the wall-clock time taken at the start of every flow step (= every call of the model
    closure).  ``max_steps``: the closure raises a private exception when step ``max_steps`` is about to start, which ends the
    reference's loop from outside -- nothing in the reference is edited, and the steps that did run are its own.
    Returns {"step_start": [t_0 .. t_k], "x_t_after_step": [x_t after step 0 .. k-1] (the inputs of the following model calls),
    "result": the full result dict when the loop ran to the end, else None}."""
    import time
    ns = load_reference()
    model = build_reference_dit(cfg, state_dict, dtype)
    cond = inputs["pointclouds"].to(dtype); feats = inputs["features"].to(dtype); scales = inputs["scales"].to(dtype)
    anchor = inputs["anchor_indices"]; ppp = inputs["points_per_part"]; x_1 = inputs["x_1"].to(dtype)
    cu_part = F.pad(torch.cumsum(ppp[ppp > 0], 0), (1, 0)).to(torch.int32)
    cu_batch = inputs["cu_seqlens"].to(torch.int32)
    B = cu_batch.shape[0] - 1
    starts, xts = [], []

    @torch.inference_mode()
    def run():
        def fn(x, t):
            starts.append(time.perf_counter())
            if len(starts) > 1:
                xts.append(x.clone())
            if max_steps is not None and len(starts) > max_steps:
                raise _StopAfter()
            ts = torch.full((B,), t, dtype=dtype)
            return model(x=x, timesteps=ts, cond_coord=cond, local_features=feats, latent_features=None, scales=scales,
                         anchor_indices=anchor, cu_seqlens_batch=cu_batch, cu_seqlens_part=cu_part)
        try:
            res = ns.get_sampler("euler")(flow_model_fn=fn, x_1=x_1, x_0=cond, condition=cond, points_per_part=ppp,
                                          cu_seqlens_batch=cu_batch, anchor_indices=anchor, num_steps=num_steps,
                                          return_trajectory=True, rigidity_forcing=rigidity_forcing)
        except _StopAfter:
            return None
        R, t = ns.fit_transformations(cond, res["end_point_trajectory"][-1], ppp, cu_batch)
        starts.append(time.perf_counter())
        return {"end_point_trajectory": res["end_point_trajectory"], "trajectory": res["trajectory"], "R": R, "t": t}
    result = run()
    return {"step_start": starts, "x_t_after_step": xts, "result": result}

