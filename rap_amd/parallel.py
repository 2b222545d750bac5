"""Multi-GPU sharding of the sampling path: independent scan pairs, no data-path exchange.

Every op on the path is segmented per sample (attention by cu_seqlens, adaLN per sample, Procrustes per
part), so samples shard embarrassingly: rank r of W owns a contiguous block of pairs, weights are
replicated, and the only collective is ONE all-gather of the registered clouds and poses after sampling
(RCCL over xGMI on the GPU box: backend "nccl"; "gloo" in the CPU tests).  The reference's own sharding is
rank-strided (data/datamodule.py:103-106) and single-GPU at inference (config/trainer/infer.yaml:3).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items: int, world_size: int, rank: int) -> range:
    """Contiguous block partition; the first (n_items % world_size) ranks get one extra item."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    q, r = divmod(n_items, world_size)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def gather_registrations(final_points: torch.Tensor, R: torch.Tensor, t: torch.Tensor, group=None):
    """All-gather equal-shaped per-rank results: (TPr,3), (Br,P,3,3), (Br,P,3) -> (W*TPr,3), (W*Br,P,3,3), (W*Br,P,3).

    One flat buffer, one collective: the payload (a few MB per rank) is latency-bound, so the three tensors
    travel together."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return final_points, R, t
    world = dist.get_world_size(group)
    flat = torch.cat([final_points.reshape(-1), R.reshape(-1), t.reshape(-1)])
    out = torch.empty(world * flat.numel(), dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(out, flat.contiguous(), group=group)
    out = out.view(world, -1)
    n0, n1 = final_points.numel(), R.numel()
    pts = out[:, :n0].reshape(world * final_points.shape[0], 3)
    Rg = out[:, n0:n0 + n1].reshape(world * R.shape[0], *R.shape[1:])
    tg = out[:, n0 + n1:].reshape(world * t.shape[0], *t.shape[1:])
    return pts, Rg, tg
