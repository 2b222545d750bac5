"""Host-side mirror of the reference's nearest-neighbour registration metrics.

``compute_cd`` and ``compute_correspondence_rmse`` keep the reference signatures and return structures
(``rectified_point_flow/eval/metrics.py:14-48`` and ``:386-469``); the N x M distance work runs in one LDS-tiled kernel
(``nn_metrics.hip``) instead of pytorch3d's chamfer op per object / a dense ``torch.cdist`` matrix.
"""
from __future__ import annotations

import torch

from . import _lib
from .flow_model import _f32c, _require_cuda, workspace


def compute_cd(pointclouds_gt, pointclouds_pred, cu_seqlens_batch, anchor_indices=None) -> torch.Tensor:
    """-> (B,) whole-object chamfer RMSE: sqrt(0.5 * (mean_i min_j |gt_i - pred_j|^2 + mean_j min_i |pred_j - gt_i|^2))."""
    gt = pointclouds_gt.reshape(-1, 3)
    pred = pointclouds_pred.reshape(-1, 3)
    _require_cuda(pred, "pointclouds_pred")
    device = pred.device
    gt, pred = _f32c(gt.to(device)), _f32c(pred)
    cu = cu_seqlens_batch.to(device=device, dtype=torch.int32).contiguous()
    B, TP = cu.shape[0] - 1, pred.shape[0]
    out = torch.empty((B,), dtype=torch.float32, device=device)
    lib = _lib.load()
    ws = workspace(device, lib.rap_nn_metrics_workspace_bytes(TP, B))
    with torch.cuda.device(device):
        rc = lib.rap_chamfer_rmse(_lib.ptr(gt), _lib.ptr(pred), _lib.ptr(cu), B, TP, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                  _lib.current_stream(device))
    _lib.check(rc, "rap_chamfer_rmse")
    return out


def compute_correspondence_rmse(source_gt, target_gt, source_pred, target_pred, distance_threshold: float = 0.1):
    """-> (rmse tensor, num_correspondences int, correspondence_ratio float), as the reference (one scan pair).  The two Python
    numbers of the return value are the reference's API and cost the one host read-back this function makes."""
    def _2d(pc, name):                                                  # metrics.py:414-423
        if pc.dim() == 1:
            pc = pc.unsqueeze(0)
        if pc.dim() == 3:
            if pc.shape[0] != 1:
                raise ValueError(f"This function only works for a single pair of point clouds. {name} has shape (B, N, 3) with B > 1.")
            pc = pc.squeeze(0)
        return pc
    sg, tg, sp, tp = (_2d(x, n) for x, n in ((source_gt, "source_gt"), (target_gt, "target_gt"), (source_pred, "source_pred"),
                                              (target_pred, "target_pred")))
    _require_cuda(sg, "source_gt")
    device = sg.device
    Ns, Nt = sg.shape[0], tg.shape[0]
    if Ns == 0 or Nt == 0:
        return torch.tensor(float("inf"), device=device), 0, 0.0          # :433-434
    if sp.shape[0] != Ns:
        raise ValueError(f"source_pred must have the same number of points as source_gt. Got {sp.shape[0]} vs {Ns}.")
    if tp.shape[0] != Nt:
        raise ValueError(f"target_pred must have the same number of points as target_gt. Got {tp.shape[0]} vs {Nt}.")
    sg, tg, sp, tp = _f32c(sg), _f32c(tg.to(device)), _f32c(sp.to(device)), _f32c(tp.to(device))
    out = torch.empty((3,), dtype=torch.float32, device=device)
    lib = _lib.load()
    ws = workspace(device, lib.rap_nn_metrics_workspace_bytes(Ns, 1))
    with torch.cuda.device(device):
        rc = lib.rap_correspondence_rmse(_lib.ptr(sg), _lib.ptr(tg), _lib.ptr(sp), _lib.ptr(tp), Ns, Nt, float(distance_threshold),
                                         _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.current_stream(device))
    _lib.check(rc, "rap_correspondence_rmse")
    host = out.cpu()
    n = int(host[1].item())
    if n == 0:
        return torch.tensor(float("inf"), device=device), 0, 0.0          # :453-454
    return out[0], n, n / Ns


def compute_transform_errors(pointclouds, pointclouds_gt, rotations_gt, translations_gt, rotations_pred, translations_pred, points_per_part,
                             anchor_part, matched_part_ids=None, scale=None, cu_seqlens_batch=None, use_icp: bool = False,
                             return_per_part: bool = False):
    """Reference signature (eval/metrics.py:165-303) -> (rot_errors_mean (B,) in degrees, trans_errors_mean (B,)); the RRE / RTE of the
    registration task.  ``use_icp=False`` only: the ICP refinement is pytorch3d's and, in the reference's own words, "for the interchangable
    case, which does not apply for point cloud registration tasks" (:264).  ``pointclouds`` / ``pointclouds_gt`` / ``cu_seqlens_batch`` are
    accepted for signature parity -- the no-ICP branch reads only the poses.  One kernel, no host synchronisation (the reference loops
    B x P with a ``.nonzero()`` per sample).  ``return_per_part`` (an extension) also returns the (B,P) per-part errors."""
    if use_icp:
        raise NotImplementedError("use_icp=True (pytorch3d iterative_closest_point) is not part of the registration path (metrics.py:264)")
    _require_cuda(rotations_pred, "rotations_pred")
    device = rotations_pred.device
    B, P = points_per_part.shape
    Rg, tg = _f32c(rotations_gt.to(device).reshape(B, P, 3, 3)), _f32c(translations_gt.to(device).reshape(B, P, 3))
    Rp, tp = _f32c(rotations_pred.reshape(B, P, 3, 3)), _f32c(translations_pred.to(device).reshape(B, P, 3))
    ppp = points_per_part.to(device=device, dtype=torch.int64).contiguous()
    anc = anchor_part.to(device=device, dtype=torch.uint8).contiguous()
    mid = None if matched_part_ids is None else matched_part_ids.to(device=device, dtype=torch.int64).contiguous()
    sc = None if scale is None else _f32c(scale.to(device).reshape(B))
    rot_pp = torch.empty((B, P), dtype=torch.float32, device=device); trans_pp = torch.empty_like(rot_pp)
    rot_m = torch.empty((B,), dtype=torch.float32, device=device); trans_m = torch.empty_like(rot_m)
    lib = _lib.load()
    with torch.cuda.device(device):
        rc = lib.rap_transform_errors(_lib.ptr(Rg), _lib.ptr(tg), _lib.ptr(Rp), _lib.ptr(tp), _lib.ptr(ppp), _lib.ptr(anc), _lib.ptr(mid),
                                      _lib.ptr(sc), B, P, _lib.ptr(rot_pp), _lib.ptr(trans_pp), _lib.ptr(rot_m), _lib.ptr(trans_m),
                                      _lib.current_stream(device))
    _lib.check(rc, "rap_transform_errors")
    if return_per_part:
        return rot_m, trans_m, rot_pp, trans_pp
    return rot_m, trans_m
