"""Input side of the sampling boundary: raw multi-part scans -> the packed batch ``RectifiedPointFlow.sample_rectified_flow``
consumes.  Device-side replacement, in their evaluation-split form (no augmentation), of the reference's numpy
``PointCloudDataset._transform`` (rectified_point_flow/data/dataset.py:733-900) and ``variable_collate_fn``
(rectified_point_flow/data/datamodule.py:169-198): same keys, same conventions (largest part = anchor, centred on its centroid and
scaled by 1.5 x its largest |coordinate|; the other parts centred on their own centroids; ``rotations`` = I,
``translations`` = part centroids in the globally centred frame, anchor: ``-gt_trans``).  The arithmetic is one call into
librapflow (``rap_collate_transform``); this module only packs pointers.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

from . import _lib
from .flow_model import _require_cuda, workspace


def draw_part_permutations(counts: Sequence[int]) -> np.ndarray:
    """The within-part shuffles of ``_proc_part`` (dataset.py:819): one ``np.random.permutation(n)`` per part, in part order, from
    numpy's global RNG like the reference -- seed it (``np.random.seed``) for reproducible batches."""
    return np.concatenate([np.random.permutation(int(n)) for n in counts if int(n) > 0]) if len(counts) else np.zeros(0, np.int64)


@torch.inference_mode()
def transform_and_collate(samples: Sequence[dict], max_parts: int, shuffle: bool = True, order: torch.Tensor | None = None,
                          device: torch.device | str | None = None) -> dict:
    """samples: list of ``{"parts": [ (n_i,3) tensors / arrays, fp32 or fp64 ], "features": [ (n_i,F) ] (optional)}``.

    Returns the collated batch dict of the reference (``pointclouds``, ``pointclouds_gt``, ``features``, ``rotations``,
    ``translations``, ``points_per_part``, ``part_indices``, ``scales``, ``anchor_parts``, ``anchor_indices``, ``init_rotation``,
    ``global_rotation``, ``global_translation``, ``cu_seqlens``, ``num_parts``), every tensor on the device.
    ``shuffle``: draw the within-part permutations with numpy's global RNG exactly as the reference does; ``order`` overrides
    them ((TP,) int64, index inside the part for every output point)."""
    if len(samples) == 0:
        raise ValueError("empty batch")
    first = samples[0]["parts"][0]
    if device is None:
        device = first.device if isinstance(first, torch.Tensor) else torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    B, P = len(samples), int(max_parts)
    counts = np.zeros((B, P), dtype=np.int64)
    pts, feats = [], []
    has_feat = all(s.get("features") is not None for s in samples)
    f64 = False
    for b, s in enumerate(samples):
        if len(s["parts"]) > P:
            raise ValueError(f"sample {b} has {len(s['parts'])} parts > max_parts={P}")
        if len(s["parts"]) == 0:
            raise ValueError(f"sample {b} has no parts")
        for i, part in enumerate(s["parts"]):
            t = part if isinstance(part, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(part))
            if t.dim() != 2 or t.shape[1] != 3 or t.shape[0] == 0:
                raise ValueError(f"sample {b} part {i}: expected a non-empty (n,3) cloud")
            f64 = f64 or t.dtype == torch.float64
            counts[b, i] = t.shape[0]
            pts.append(t)
            if has_feat:
                f = s["features"][i]
                feats.append(f if isinstance(f, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(f)))
    dt = torch.float64 if f64 else torch.float32
    points = torch.cat([t.to(device=device, dtype=dt) for t in pts]).contiguous()
    _require_cuda(points, "parts")
    TP = points.shape[0]
    feat_in = torch.cat([f.to(device=device, dtype=torch.float32) for f in feats]).contiguous() if has_feat else None
    F = feat_in.shape[1] if has_feat else 0
    if order is None and shuffle:
        order = torch.from_numpy(draw_part_permutations(counts.reshape(-1)))
    if order is not None:
        order = order.to(device=device, dtype=torch.int64).contiguous()
        if order.shape != (TP,):
            raise ValueError(f"order must have shape ({TP},)")
    ppp = torch.from_numpy(counts).to(device)
    out = {
        "pointclouds": torch.empty(TP, 3, device=device), "pointclouds_gt": torch.empty(TP, 3, device=device),
        "anchor_indices": torch.empty(TP, dtype=torch.uint8, device=device), "part_indices": torch.empty(TP, dtype=torch.int64, device=device),
        "rotations": torch.empty(B, P, 3, 3, device=device), "translations": torch.empty(B, P, 3, device=device),
        "scales": torch.empty(B, device=device), "anchor_parts": torch.empty(B, P, dtype=torch.uint8, device=device),
        "global_translation": torch.empty(B, 3, device=device), "cu_seqlens": torch.empty(B + 1, dtype=torch.int64, device=device),
    }
    feat_out = torch.empty(TP, F, device=device) if has_feat else None
    flag = torch.zeros(1, dtype=torch.int32, device=device) if order is not None else None
    lib = _lib.load()
    ws = workspace(device, lib.rap_collate_workspace_bytes(B, P))
    rc = lib.rap_collate_transform(_lib.ptr(points), 1 if f64 else 0, _lib.ptr(ppp), B, P, TP, _lib.ptr(order), _lib.ptr(feat_in), F,
                                   _lib.ptr(out["pointclouds"]), _lib.ptr(out["pointclouds_gt"]), _lib.ptr(feat_out),
                                   _lib.ptr(out["anchor_indices"]), _lib.ptr(out["part_indices"]), _lib.ptr(out["rotations"]),
                                   _lib.ptr(out["translations"]), _lib.ptr(out["scales"]), _lib.ptr(out["anchor_parts"]),
                                   _lib.ptr(out["global_translation"]), _lib.ptr(out["cu_seqlens"]), _lib.ptr(flag), _lib.ptr(ws), ws.numel(),
                                   _lib.current_stream(device))
    _lib.check(rc, "rap_collate_transform")
    if flag is not None and int(flag.item()) != 0:          # only when a caller-supplied order is used: one 4-byte read
        raise ValueError("order holds an index outside its part")
    out["anchor_indices"] = out["anchor_indices"].bool()
    out["anchor_parts"] = out["anchor_parts"].bool()
    out["points_per_part"] = ppp
    if has_feat:
        out["features"] = feat_out
    eye = torch.eye(3, device=device).expand(B, 3, 3).contiguous()
    out["init_rotation"] = eye
    out["global_rotation"] = eye.clone()
    out["num_parts"] = [len(s["parts"]) for s in samples]
    return out
