"""Host-side mirror of ``rectified_point_flow/procrustes.py`` over librapflow's device kernels.

Same names, argument meaning and error behaviour as the reference
(``fit_transformations`` procrustes.py:40-84, ``rigidify_prediction_with_procrustes`` :86-118,
``solve_procrustes`` :6-37); the Python double loop with one SVD launch and >= 2 host syncs per
part is replaced by three kernels per call and no sync.
"""
from __future__ import annotations

import torch

from . import _lib
from .flow_model import _f32c, _require_cuda, workspace


def _check_packed(pcd: torch.Tensor, points_per_part: torch.Tensor, cu_seqlens_batch):
    if pcd.ndim == 2 and pcd.shape[1] == 3:
        if cu_seqlens_batch is None:   # utils/point_clouds.py:28-29
            raise ValueError("cu_seqlens_batch is required when pointclouds has shape (TP, 3)")
    elif pcd.ndim == 3:
        pcd = pcd.reshape(-1, 3)       # fixed batching (B,N,3): flat order is already (b, p)-major
    else:
        raise ValueError(f"unsupported point cloud shape {tuple(pcd.shape)}")
    return pcd


def fit_transformations(source_pcds, target_pcds, points_per_part, cu_seqlens_batch=None):
    """-> R (B,P,3,3), t (B,P,3) with target ~= source @ R^T + t; zero rows for empty parts."""
    src = _check_packed(source_pcds, points_per_part, cu_seqlens_batch)
    tgt = _check_packed(target_pcds, points_per_part, cu_seqlens_batch)
    _require_cuda(src, "source_pcds")
    device = src.device
    B, P = points_per_part.shape
    src, tgt = _f32c(src), _f32c(tgt)
    ppp = points_per_part.to(device=device, dtype=torch.int64).contiguous()
    lib = _lib.load()
    R = torch.empty((B, P, 3, 3), dtype=torch.float32, device=device)
    t = torch.empty((B, P, 3), dtype=torch.float32, device=device)
    ws = workspace(device, lib.rap_procrustes_workspace_bytes(B * P))
    with torch.cuda.device(device):
        rc = lib.rap_fit_transformations(_lib.ptr(src), _lib.ptr(tgt), _lib.ptr(ppp), B, P, _lib.ptr(R), _lib.ptr(t),
                                         _lib.ptr(ws), ws.numel(), _lib.current_stream(device))
    _lib.check(rc, "rap_fit_transformations")
    return R, t


def rigidify_prediction_with_procrustes(prediction, condition, points_per_part, cu_seqlens_batch=None):
    """-> (TP,3): every part of ``condition`` moved by its best rigid fit onto ``prediction``."""
    pred = _check_packed(prediction, points_per_part, cu_seqlens_batch)
    cond = _check_packed(condition, points_per_part, cu_seqlens_batch)
    _require_cuda(pred, "prediction")
    device = pred.device
    B, P = points_per_part.shape
    pred, cond = _f32c(pred), _f32c(cond)
    ppp = points_per_part.to(device=device, dtype=torch.int64).contiguous()
    lib = _lib.load()
    out = torch.zeros_like(pred)       # procrustes.py:103 (rows not covered by any part stay zero)
    ws = workspace(device, lib.rap_procrustes_workspace_bytes(B * P))
    with torch.cuda.device(device):
        rc = lib.rap_rigidify(_lib.ptr(pred), _lib.ptr(cond), _lib.ptr(ppp), B, P, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                              _lib.current_stream(device))
    _lib.check(rc, "rap_rigidify")
    return out.reshape(prediction.shape)


def rigidify_blend(x0_hat, condition, points_per_part, x_1, w0: float, w1: float):
    """x_t = rigidify(x0_hat, cond) * w0 + x_1 * w1  (sampler.py:59-60) in one launch sequence."""
    _require_cuda(x0_hat, "x0_hat")
    device = x0_hat.device
    B, P = points_per_part.shape
    x0, cond, x1 = _f32c(x0_hat.reshape(-1, 3)), _f32c(condition.reshape(-1, 3)), _f32c(x_1.reshape(-1, 3))
    ppp = points_per_part.to(device=device, dtype=torch.int64).contiguous()
    lib = _lib.load()
    out = torch.zeros_like(x0)
    ws = workspace(device, lib.rap_procrustes_workspace_bytes(B * P))
    with torch.cuda.device(device):
        rc = lib.rap_rigidify_blend(_lib.ptr(x0), _lib.ptr(cond), _lib.ptr(ppp), B, P, _lib.ptr(x1), float(w0), float(w1),
                                    _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.current_stream(device))
    _lib.check(rc, "rap_rigidify_blend")
    return out.reshape(x0_hat.shape)


def solve_procrustes(source_pcd: torch.Tensor, target_pcd: torch.Tensor):
    """Single pair (N,3),(N,3) -> R (3,3), t (3,)  (procrustes.py:6-37)."""
    n = source_pcd.shape[0]
    ppp = torch.tensor([[n]], dtype=torch.int64)
    cu = torch.tensor([0, n], dtype=torch.int32)
    R, t = fit_transformations(source_pcd, target_pcd, ppp, cu)
    return R[0, 0], t[0, 0]
