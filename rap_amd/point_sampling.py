"""Host-side mirror of the reference's batched farthest point sampling (preprocessing in front of
MiniSpinNet): ``apply_batched_fps`` of ``dataset_process/utils/point_sampling_utils.py:263-305``, i.e.
``pytorch3d.ops.sample_farthest_points(points, lengths=..., K=..., random_start_point=True)`` after ``torch.manual_seed``.
Also ``voxel_down_sample_torch`` of ``dataset_process/utils/dataset_utils.py:279-322`` (the down-sampling step before it).
"""
from __future__ import annotations

import os

import torch

from . import _lib
from .flow_model import _f32c, _require_cuda


def sample_farthest_points(points: torch.Tensor, lengths: torch.Tensor | None = None, K=50, random_start_point: bool = False,
                           start_idx: torch.Tensor | None = None):
    """pytorch3d signature: points (N,P,3) zero-padded, lengths (N,), K int or (N,) -> (sampled (N,Kmax,3), idx (N,Kmax) int64),
    padded with 0 / -1 past min(K_n, length_n).  The random start (one ``torch.randint(high=length_n)`` per cloud, in cloud order,
    as pytorch3d draws it) can be overridden with ``start_idx``."""
    _require_cuda(points, "points")
    device = points.device
    N, P, _ = points.shape
    lengths = torch.full((N,), P, dtype=torch.int64) if lengths is None else lengths.to("cpu", torch.int64)
    Ks = torch.full((N,), int(K), dtype=torch.int64) if isinstance(K, int) else K.to("cpu", torch.int64)
    Kmax = int(Ks.max())
    if start_idx is None:
        if random_start_point:
            start_idx = torch.tensor([int(torch.randint(high=int(lengths[n]), size=(1,)).item()) for n in range(N)])
        else:
            start_idx = torch.zeros(N, dtype=torch.int64)
    pts = _f32c(points)                                  # padded layout: cloud n occupies rows [n*P, n*P + length_n)
    i32 = lambda t: t.to(device=device, dtype=torch.int32).contiguous()
    cloud_start = i32(torch.arange(N, dtype=torch.int64) * P)
    idx_out = torch.empty((N, Kmax), dtype=torch.int32, device=device)
    dist = torch.empty((N * P,), dtype=torch.float32, device=device)
    lib = _lib.load()
    len_d, k_d, st_d = i32(lengths), i32(Ks), i32(start_idx)      # named: a temporary's memory is recycled by the next allocation
    with torch.cuda.device(device):
        rc = lib.rap_farthest_point_sampling(_lib.ptr(pts), _lib.ptr(cloud_start), _lib.ptr(len_d), _lib.ptr(k_d), _lib.ptr(st_d), N,
                                             Kmax, _lib.ptr(idx_out), _lib.ptr(dist), _lib.current_stream(device))
    _lib.check(rc, "rap_farthest_point_sampling")
    idx = idx_out.to(torch.int64)
    gathered = torch.gather(pts, 1, idx.clamp_min(0)[..., None].expand(-1, -1, 3))         # data movement only
    sampled = torch.where((idx >= 0)[..., None], gathered, torch.zeros_like(gathered))
    return sampled, idx


def apply_batched_fps(batch_augmented_tensor, batch_lengths_tensor, batch_k_tensor, global_seed: int, device):
    """point_sampling_utils.py:263-305 -> (list of sampled parts (K_i,3), indices (N,Kmax))."""
    torch.manual_seed(global_seed)
    batch = batch_augmented_tensor.contiguous().to(device)
    _, idx = sample_farthest_points(batch, lengths=batch_lengths_tensor, K=batch_k_tensor, random_start_point=True)
    parts = [batch[i][idx[i][: int(k)]] for i, k in enumerate(batch_k_tensor)]
    return parts, idx


def _use_sorted_voxel_path(slots: int, N: int, path: str | None) -> bool:
    """Dense key table (volume of the grid) or radix sort (O(N))?  The table is a memset + two passes over `slots` 8-byte entries:
    worth it while it is small (<= 2^27 slots = 1 GiB) or within 32x the point count; beyond that -- and beyond its hard limit -- sort.
    ``path`` / RAP_VOXEL_PATH = "dense" | "sorted" forces one (tests)."""
    path = path or os.environ.get("RAP_VOXEL_PATH")
    if path == "sorted":
        return True
    if path == "dense":
        return False
    return slots < 0 or (slots > (1 << 27) and slots > 32 * N)


def voxel_down_sample_torch(points: torch.Tensor, voxel_size: float, path: str | None = None) -> torch.Tensor:
    """dataset_utils.py:279-322: points (N,3) on the GPU -> indices (M,) int64 of the point closest to each occupied voxel's
    centre, in ascending voxel-key order (``points[indices]`` is the down-sampled cloud).  Same result as the reference's CPU path
    bit for bit (its CUDA path divides by multiplying with 1/voxel_size and reduces with a non-deterministic scatter).
    Two device paths with identical results: a dense key table over the grid volume, or -- for grids that are large against the
    number of points -- a radix sort of per-point keys (memory O(N); see ``_use_sorted_voxel_path``)."""
    _require_cuda(points, "points")
    device = points.device
    pts = _f32c(points)
    N = pts.shape[0]
    lib = _lib.load()
    bounds = torch.empty(6, dtype=torch.int64, device=device)
    dmax = torch.empty(1, dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        stream = _lib.current_stream(device)
        _lib.check(lib.rap_voxel_bounds(_lib.ptr(pts), N, float(voxel_size), _lib.ptr(bounds), _lib.ptr(dmax), stream), "rap_voxel_bounds")
        h_bounds = bounds.cpu()                            # sizes the key table (the reference synchronises here too: .item())
        h_dmax = float(dmax.cpu())
        slots = lib.rap_voxel_table_slots(h_bounds.data_ptr())
        count = torch.empty(1, dtype=torch.int32, device=device)
        if _use_sorted_voxel_path(slots, N, path):
            if int((h_bounds[3:] - h_bounds[:3]).max()) >= (1 << 18):
                raise ValueError(f"voxel grid {tuple((h_bounds[3:] - h_bounds[:3]).tolist())}: more than 2^18 cells along an axis")
            ws = torch.empty(lib.rap_voxel_sorted_workspace_bytes(N), dtype=torch.uint8, device=device)
            idx = torch.empty(N, dtype=torch.int64, device=device)
            rc = lib.rap_voxel_downsample_sorted(_lib.ptr(pts), N, float(voxel_size), h_bounds.data_ptr(), h_dmax, _lib.ptr(idx),
                                                 _lib.ptr(count), _lib.ptr(ws), ws.numel(), stream)
            _lib.check(rc, "rap_voxel_downsample_sorted")
        else:
            if slots < 0:
                raise ValueError(f"voxel grid {tuple((h_bounds[3:] - h_bounds[:3]).tolist())} needs more than 2^33 table slots")
            ws = torch.empty(lib.rap_voxel_workspace_bytes(h_bounds.data_ptr()), dtype=torch.uint8, device=device)
            idx = torch.empty(min(N, slots), dtype=torch.int64, device=device)
            rc = lib.rap_voxel_downsample(_lib.ptr(pts), N, float(voxel_size), h_bounds.data_ptr(), h_dmax, _lib.ptr(idx), _lib.ptr(count),
                                          _lib.ptr(ws), ws.numel(), stream)
            _lib.check(rc, "rap_voxel_downsample")
    return idx[: int(count.cpu())]


def remove_statistical_outlier(points: torch.Tensor, nb_neighbors: int = 20, std_ratio: float = 2.5):
    """Open3D ``PointCloud.remove_statistical_outlier(nb_neighbors, std_ratio)`` as extract_sample_features.py:378-385 calls it:
    returns ``(points[inlier_indices], inlier_indices)`` -- the filtered cloud and the ascending int64 indices of the kept points.
    Rule (Open3D's published algorithm; the wheel is not in the reference mount: parity unpinned): a point is kept iff the mean
    distance d to its ``nb_neighbors`` nearest points (itself included) satisfies 0 < d < mean(d) + std_ratio * std(d)."""
    _require_cuda(points, "points")
    device = points.device
    pts = _f32c(points)
    N = pts.shape[0]
    if N == 0:
        return pts, torch.zeros(0, dtype=torch.int64, device=device)
    lib = _lib.load()
    idx = torch.empty(N, dtype=torch.int64, device=device)
    count = torch.empty(1, dtype=torch.int32, device=device)
    ws = torch.empty(lib.rap_outlier_workspace_bytes(N), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        rc = lib.rap_statistical_outliers(_lib.ptr(pts), N, int(nb_neighbors), float(std_ratio), _lib.ptr(idx), _lib.ptr(count), _lib.ptr(None),
                                          _lib.ptr(ws), ws.numel(), _lib.current_stream(device))
    _lib.check(rc, "rap_statistical_outliers")
    idx = idx[: int(count.cpu())]                       # the caller needs the size (Open3D returns a list)
    return pts[idx], idx


def calculate_voxel_coverage(points: torch.Tensor, voxel_size: float, path: str | None = None) -> int:
    """point_sampling_utils.py:11-31: number of distinct voxels floor(p / voxel_size) the cloud occupies (exact key: the
    down-sampling table reproduces the reference's colliding cubic key and cannot be used to count).  A byte table over the grid's
    bounding box, or a radix sort of the per-point voxel ids when that box is large against the number of points.

    Precision: the cloud is taken as float32 and the voxel index is floor(fp32(p) / fp32(voxel_size)).  The reference computes
    np.floor(points / voxel_size) in the array's OWN dtype, so for float64 input a point within one fp32 rounding of a voxel face
    (|p / voxel_size - round(p / voxel_size)| < ~6e-8 |p / voxel_size|) can be counted in the neighbouring voxel; identical counts are
    guaranteed (and tested against the reference's own function) for float32 clouds, which is what the reference's pipeline feeds it
    (extract_sample_features.py loads .ply vertices as float32).  The same holds for voxel_down_sample_torch and
    remove_statistical_outlier below."""
    if points.shape[0] == 0:
        return 0
    _require_cuda(points, "points")
    device = points.device
    pts = _f32c(points)
    N = pts.shape[0]
    lib = _lib.load()
    bounds = torch.empty(6, dtype=torch.int64, device=device)
    dmax = torch.empty(1, dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        stream = _lib.current_stream(device)
        _lib.check(lib.rap_voxel_bounds(_lib.ptr(pts), N, float(voxel_size), _lib.ptr(bounds), _lib.ptr(dmax), stream), "rap_voxel_bounds")
        h_bounds = bounds.cpu()
        nbytes = lib.rap_voxel_coverage_workspace_bytes(h_bounds.data_ptr())
        count = torch.empty(1, dtype=torch.int64, device=device)
        path = path or os.environ.get("RAP_VOXEL_PATH")
        if path == "sorted" or (path != "dense" and (nbytes == 0 or (nbytes > (1 << 30) and nbytes > 256 * N))):
            ws = torch.empty(lib.rap_voxel_sorted_workspace_bytes(N), dtype=torch.uint8, device=device)
            rc = lib.rap_voxel_coverage_sorted(_lib.ptr(pts), N, float(voxel_size), h_bounds.data_ptr(), _lib.ptr(count), _lib.ptr(ws),
                                               ws.numel(), stream)
            _lib.check(rc, "rap_voxel_coverage_sorted")
        else:
            if nbytes == 0:
                raise ValueError(f"voxel grid {tuple((h_bounds[3:] - h_bounds[:3] + 1).tolist())} is too large for the coverage table")
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            rc = lib.rap_voxel_coverage(_lib.ptr(pts), N, float(voxel_size), h_bounds.data_ptr(), _lib.ptr(count), _lib.ptr(ws), ws.numel(), stream)
            _lib.check(rc, "rap_voxel_coverage")
    return int(count.cpu())


def calculate_adaptive_sample_count_per_part(parts_points, voxel_size: float, voxel_ratio: float, min_points_per_part: int,
                                             max_sample_points: int) -> list[int]:
    """point_sampling_utils.py:33-84: per part min(max(min_points_per_part, int(occupied_voxels * voxel_ratio)), n_points,
    max_sample_points); 0 for an empty part."""
    out = []
    for p in parts_points:
        n = int(p.shape[0])
        if n == 0:
            out.append(0)
            continue
        c = int(calculate_voxel_coverage(p, voxel_size) * voxel_ratio)
        out.append(min(max_sample_points, min(n, max(min_points_per_part, c))))
    return out
