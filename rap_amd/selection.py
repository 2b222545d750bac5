"""Host-side mirror of the reference's generation selection by rigidity.

``compute_rigidity_rmse`` keeps the reference signature (``rectified_point_flow/eval/metrics.py:511-622``);
``average_trajectory_rigidity_rmse`` and ``select_generations_by_rigidity`` are the two inner blocks of
``RectifiedPointFlow.test_step`` (``modeling.py:466-500`` and ``:518, 560-592``), which the reference runs as a
``generations x steps x B x P`` Python loop with a host sync per part.  Here each is one call into librapflow and
nothing syncs with the host.
"""
from __future__ import annotations

import torch

from . import _lib
from .flow_model import _f32c, _require_cuda, workspace
from .procrustes import _check_packed


def compute_rigidity_rmse(pointclouds_input, pointclouds_pred, rotations_pred, translations_pred, points_per_part,
                          cu_seqlens_batch=None, scales=None, average_per_part: bool = False) -> torch.Tensor:
    """-> (B,) per-object rigidity RMSE (metres if ``scales`` is given), ``inf`` for an object without points."""
    src = _check_packed(pointclouds_input, points_per_part, cu_seqlens_batch)
    pred = _check_packed(pointclouds_pred, points_per_part, cu_seqlens_batch)
    _require_cuda(src, "pointclouds_input")
    device = src.device
    B, P = points_per_part.shape
    src, pred = _f32c(src), _f32c(pred)
    R, t = _f32c(rotations_pred.to(device)), _f32c(translations_pred.to(device))
    if tuple(R.shape) != (B, P, 3, 3) or tuple(t.shape) != (B, P, 3):
        raise ValueError("rotations_pred / translations_pred must be (B,P,3,3) / (B,P,3)")
    ppp = points_per_part.to(device=device, dtype=torch.int64).contiguous()
    sc = None if scales is None else _f32c(scales.to(device))
    lib = _lib.load()
    out = torch.empty((B,), dtype=torch.float32, device=device)
    ws = workspace(device, lib.rap_rigidity_workspace_bytes(B * P, 0, B))
    with torch.cuda.device(device):
        rc = lib.rap_rigidity_rmse(_lib.ptr(src), _lib.ptr(pred), _lib.ptr(R), _lib.ptr(t), _lib.ptr(ppp), B, P, _lib.ptr(sc),
                                   1 if average_per_part else 0, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                   _lib.current_stream(device))
    _lib.check(rc, "rap_rigidity_rmse")
    return out


def average_trajectory_rigidity_rmse(condition, trajectory, points_per_part, cu_seqlens_batch=None, scales=None,
                                     return_per_step: bool = False):
    """Mean over the steps of an end-point trajectory (S,TP,3) of the rigidity RMSE of x0_hat(step) against its own
    Procrustes fit from ``condition`` (modeling.py:466-489) -> (B,) [, (S,B)]."""
    cond = _check_packed(condition, points_per_part, cu_seqlens_batch)
    _require_cuda(cond, "condition")
    device = cond.device
    B, P = points_per_part.shape
    cond = _f32c(cond)
    traj = _f32c(trajectory.to(device))
    S = traj.shape[0]
    TP = cond.shape[0]
    if traj.numel() != S * TP * 3:
        raise ValueError("trajectory must be (steps, TP, 3)")
    ppp = points_per_part.to(device=device, dtype=torch.int64).contiguous()
    sc = None if scales is None else _f32c(scales.to(device))
    lib = _lib.load()
    mean = torch.empty((B,), dtype=torch.float32, device=device)
    per_step = torch.empty((S, B), dtype=torch.float32, device=device) if return_per_step else None
    ws = workspace(device, lib.rap_rigidity_workspace_bytes(B * P, S, B))
    with torch.cuda.device(device):
        rc = lib.rap_trajectory_rigidity_rmse(_lib.ptr(cond), _lib.ptr(traj), _lib.ptr(ppp), B, P, TP, S, _lib.ptr(sc),
                                              _lib.ptr(mean), _lib.ptr(per_step), _lib.ptr(ws), ws.numel(),
                                              _lib.current_stream(device))
    _lib.check(rc, "rap_trajectory_rigidity_rmse")
    return (mean, per_step) if return_per_step else mean


def compute_overlap_ratio(pointclouds_pred, points_per_part, cu_seqlens_batch=None, taus=(0.005, 0.01, 0.02),
                          return_min_distances: bool = False):
    """Reference signature (eval/metrics.py:625-631) -> (T,B) overlap ratios: the fraction of an object's points that have a
    point of a DIFFERENT part within ``tau`` [, (TP,) distance to the nearest other-part point]."""
    pts = _check_packed(pointclouds_pred, points_per_part, cu_seqlens_batch)
    _require_cuda(pts, "pointclouds_pred")
    device = pts.device
    B, P = points_per_part.shape
    pts = _f32c(pts)
    TP = pts.shape[0]
    ppp = points_per_part.to(device=device, dtype=torch.int64).contiguous()
    if cu_seqlens_batch is None:     # fixed batching (B,N,3): every object has N points
        cu = torch.arange(0, TP + 1, TP // B, dtype=torch.int32, device=device)
    else:
        cu = cu_seqlens_batch.to(device=device, dtype=torch.int32).contiguous()
    tau_list = [float(taus)] if isinstance(taus, (float, int)) else [float(t) for t in taus]     # metrics.py:648-651
    if not 1 <= len(tau_list) <= 8:
        raise ValueError("between 1 and 8 thresholds are supported")
    import ctypes
    h_taus = (ctypes.c_float * len(tau_list))(*tau_list)
    lib = _lib.load()
    ratios = torch.empty((len(tau_list), B), dtype=torch.float32, device=device)
    min_d = torch.empty((TP,), dtype=torch.float32, device=device) if return_min_distances else None
    ws = workspace(device, lib.rap_overlap_workspace_bytes(TP, B, P))
    with torch.cuda.device(device):
        rc = lib.rap_overlap_ratio(_lib.ptr(pts), _lib.ptr(ppp), _lib.ptr(cu), B, P, TP, ctypes.cast(h_taus, ctypes.c_void_p),
                                   len(tau_list), _lib.ptr(ratios), _lib.ptr(min_d), _lib.ptr(ws), ws.numel(),
                                   _lib.current_stream(device))
    _lib.check(rc, "rap_overlap_ratio")
    return (ratios, min_d) if return_min_distances else ratios


def select_generations_by_overlap(stacked_overlap_ratios, final_clouds, rotations, translations, cu_seqlens_batch):
    """Per object the generation with the LARGEST overlap ratio (modeling.py:597-601) -> (best, cloud, R, t)."""
    return select_generations_by_rigidity(stacked_overlap_ratios, final_clouds, rotations, translations, cu_seqlens_batch,
                                          _pick_largest=True)


def select_generations_by_rigidity(stacked_rigidity, final_clouds, rotations, translations, cu_seqlens_batch,
                                   _pick_largest: bool = False):
    """stacked_rigidity (G,B); final_clouds (G,TP,3); rotations (G,B,P,3,3); translations (G,B,P,3) ->
    (best_gen_indices (B,) int64, cloud (TP,3), R (B,P,3,3), t (B,P,3)) of the generation with the smallest rigidity
    RMSE per object (modeling.py:518, 560-592)."""
    _require_cuda(stacked_rigidity, "stacked_rigidity")
    device = stacked_rigidity.device
    rm = _f32c(stacked_rigidity)
    G, B = rm.shape
    clouds = _f32c(final_clouds.to(device)); R = _f32c(rotations.to(device)); t = _f32c(translations.to(device))
    TP = clouds.shape[1]
    P = R.shape[2]
    if tuple(clouds.shape) != (G, TP, 3) or tuple(R.shape) != (G, B, P, 3, 3) or tuple(t.shape) != (G, B, P, 3):
        raise ValueError("shape mismatch between generations")
    cu = cu_seqlens_batch.to(device=device, dtype=torch.int32).contiguous()
    lib = _lib.load()
    best = torch.empty((B,), dtype=torch.int32, device=device)
    cloud_out = torch.empty((TP, 3), dtype=torch.float32, device=device)
    R_out = torch.empty((B, P, 3, 3), dtype=torch.float32, device=device)
    t_out = torch.empty((B, P, 3), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        rc = lib.rap_select_generation(_lib.ptr(rm), G, B, P, TP, _lib.ptr(cu), _lib.ptr(clouds), _lib.ptr(R), _lib.ptr(t),
                                       1 if _pick_largest else 0, _lib.ptr(best), _lib.ptr(cloud_out), _lib.ptr(R_out), _lib.ptr(t_out),
                                       _lib.current_stream(device))
    _lib.check(rc, "rap_select_generation")
    return best.to(torch.int64), cloud_out, R_out, t_out
