"""Host-side mirror of the reference's transform writer.

``save_transformation_files`` writes the same ``*_transform.txt`` files, names and number format as
``Evaluator._save_transformation_files`` (``rectified_point_flow/eval/evaluator.py:383-490``; consumed by
``demo.py:1332-1342``); the 4x4 matrices of a whole batch come from one kernel (``rap_relative_transforms``) instead of a
per-part numpy loop, and one device-to-host copy.
"""
from __future__ import annotations

from pathlib import Path

import torch

from . import _lib
from .flow_model import _f32c, _require_cuda


def compute_relative_transforms(rotations_pred, translations_pred, rotations_gt, translations_gt, scales, points_per_part,
                                global_rotation=None, global_translation=None) -> torch.Tensor:
    """-> (B,P,4,4) fp32: predicted pose relative to the GT pose in metres (x inv(global frame) if given); zero blocks for
    parts without points."""
    _require_cuda(rotations_pred, "rotations_pred")
    device = rotations_pred.device
    B, P = points_per_part.shape
    Rp, tp = _f32c(rotations_pred), _f32c(translations_pred.to(device))
    Rg, tg = _f32c(rotations_gt.to(device)), _f32c(translations_gt.to(device))
    sc = _f32c(scales.to(device))
    ppp = points_per_part.to(device=device, dtype=torch.int64).contiguous()
    if (global_rotation is None) != (global_translation is None):
        raise ValueError("global_rotation and global_translation must be given together")
    Gr = None if global_rotation is None else _f32c(global_rotation.to(device)).reshape(B, 3, 3)
    Gt = None if global_translation is None else _f32c(global_translation.to(device)).reshape(B, 3)
    out = torch.empty((B, P, 4, 4), dtype=torch.float32, device=device)
    lib = _lib.load()
    with torch.cuda.device(device):
        rc = lib.rap_relative_transforms(_lib.ptr(Rp), _lib.ptr(tp), _lib.ptr(Rg), _lib.ptr(tg), _lib.ptr(sc), _lib.ptr(ppp), B, P,
                                         _lib.ptr(Gr), _lib.ptr(Gt), _lib.ptr(out), _lib.current_stream(device))
    _lib.check(rc, "rap_relative_transforms")
    return out


def _suffix(generation_idx) -> str:
    if isinstance(generation_idx, str):                                     # evaluator.py:427-432
        return generation_idx if generation_idx.startswith("generation") else f"generation_{generation_idx}"
    return f"generation{generation_idx:02d}"                                # :434


def save_transformation_files(data: dict, sample_dir, dataset_name: str, sample_indices, generation_idx, rotations_pred,
                              translations_pred, global_rotation=None, global_translation=None) -> list[Path]:
    """Write ``{dataset}_sample{idx:05d}_{suffix}_part{pid:02d}_transform.txt`` for every non-empty part of every sample of
    the batch (4 rows of ``%12.8f``, evaluator.py:476-483).  ``data`` needs "rotations", "translations", "scales",
    "points_per_part" (the reference's batch schema); ``sample_indices[b]`` is the dataset index of batch element b.
    Returns the paths written."""
    ppp = data["points_per_part"]
    M = compute_relative_transforms(rotations_pred, translations_pred, data["rotations"], data["translations"], data["scales"], ppp,
                                    global_rotation, global_translation).cpu()
    ppp = ppp.cpu()
    sample_dir = Path(sample_dir)
    sample_dir.mkdir(parents=True, exist_ok=True)
    suffix = _suffix(generation_idx)
    written = []
    for b in range(ppp.shape[0]):
        for pid in torch.where(ppp[b] > 0)[0].tolist():
            path = sample_dir / f"{dataset_name}_sample{int(sample_indices[b]):05d}_{suffix}_part{pid:02d}_transform.txt"
            with open(path, "w") as f:
                for row in M[b, pid].tolist():
                    f.write(" ".join(f"{val:12.8f}" for val in row) + "\n")
            written.append(path)
    return written
