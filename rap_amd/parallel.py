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


def gather_registrations(final_points: torch.Tensor, R: torch.Tensor, t: torch.Tensor, group=None, equal_shapes: bool = False):
    """All-gather per-rank results (TPr,3), (Br,P,3,3), (Br,P,3) -> (sum TPr,3), (sum Br,P,3,3), (sum Br,P,3) in rank order.

    One flat buffer, one data collective: the payload (a few MB per rank) is latency-bound, so the three tensors travel
    together.  Ranks may hold DIFFERENT numbers of points and samples (`shard_range` hands the first ranks one more pair, and
    real scans are ragged): a 3-integer all-gather of (TPr, Br, P) precedes the data collective, every rank pads its buffer
    to the largest payload and the padding is sliced away after the gather.  `equal_shapes=True` skips that exchange (and the
    host read it needs): the caller guarantees identical shapes on every rank -- bench.py's fixed synthetic batch."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return final_points, R, t
    world = dist.get_world_size(group)
    P = R.shape[1]
    if R.shape[0] != t.shape[0] or t.shape[1] != P or final_points.dim() != 2 or final_points.shape[1] != 3:
        raise ValueError("expected final_points (TPr,3), R (Br,P,3,3), t (Br,P,3)")
    dev = final_points.device
    if equal_shapes:
        tp_all, b_all = [final_points.shape[0]] * world, [R.shape[0]] * world
    else:
        mine = torch.tensor([final_points.shape[0], R.shape[0], P], dtype=torch.int64, device=dev)
        sizes = torch.empty(world * 3, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(sizes, mine, group=group)
        sizes = sizes.view(world, 3).cpu()
        if not bool((sizes[:, 2] == P).all()):
            raise ValueError(f"ranks disagree on the number of parts per sample: {sizes[:, 2].tolist()}")
        tp_all, b_all = sizes[:, 0].tolist(), sizes[:, 1].tolist()
    n_max = max(tp * 3 + b * P * 12 for tp, b in zip(tp_all, b_all))
    n0, n1, n2 = final_points.numel(), R.numel(), t.numel()
    if n0 + n1 + n2 == n_max:
        flat = torch.cat([final_points.reshape(-1), R.reshape(-1), t.reshape(-1)])
    else:
        flat = torch.zeros(n_max, dtype=final_points.dtype, device=dev)
        flat[:n0] = final_points.reshape(-1)
        flat[n0:n0 + n1] = R.reshape(-1)
        flat[n0 + n1:n0 + n1 + n2] = t.reshape(-1)
    out = torch.empty(world * n_max, dtype=flat.dtype, device=dev)
    dist.all_gather_into_tensor(out, flat.contiguous(), group=group)
    out = out.view(world, n_max)
    pts, Rs, ts = [], [], []
    for r in range(world):
        tp, b = int(tp_all[r]), int(b_all[r])
        a0 = tp * 3; a1 = a0 + b * P * 9
        pts.append(out[r, :a0].reshape(tp, 3))
        Rs.append(out[r, a0:a1].reshape(b, P, 3, 3))
        ts.append(out[r, a1:a1 + b * P * 3].reshape(b, P, 3))
    return torch.cat(pts), torch.cat(Rs), torch.cat(ts)
