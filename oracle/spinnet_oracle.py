"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's MiniSpinNet forward (inference, global-z mode).

Follows dataset_process/utils/spinnet/patch_embedder.py:49-183, patchnet.py:49-84 and utils/common.py:213-275, 338-372,
387-469 step by step in functional torch (fp32).  pytorch3d 0.7.8 (install.sh:16) is absent: `ball_query_first_k` restates its
documented semantics (first K points of p2, in index order, with squared distance < radius^2; idx padded with -1, neighbours
with zeros) -- parity unpinned for that one function, everything else is pinned to the reference's own modules run in this
container (tests/test_oracle.py::test_spinnet_oracle_matches_live_reference).  Only tests / smoke may import this module.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def ball_query_first_k(p1, p2, K, radius):
    """p1 (P1,3) queries, p2 (P2,3) -> idx (P1,K) long (-1 padded), nn (P1,K,3) (zero padded)"""
    d2 = ((p1[:, None, :] - p2[None, :, :]) ** 2).sum(-1)
    within = d2 < radius * radius
    rank = within.long().cumsum(-1) - 1
    idx = torch.full((p1.shape[0], K), -1, dtype=torch.long)
    i, j = torch.nonzero(within & (rank < K), as_tuple=True)
    idx[i, rank[i, j]] = j
    nn = torch.zeros(p1.shape[0], K, 3, dtype=p2.dtype)
    ii, kk = torch.nonzero(idx >= 0, as_tuple=True)
    nn[ii, kk] = p2[idx[ii, kk]]
    return idx, nn


def voxel_centres(rad_n=3, azi_n=20, ele_n=7):
    """get_voxel_coordinate(radius=1, ...) (common.py:387-393 with s2_grid :213-225 and change_coordinates :355-372)"""
    beta = np.linspace(0, np.pi, ele_n, endpoint=False) + np.pi / ele_n / 2
    alpha = np.linspace(0, 2 * np.pi, azi_n, endpoint=False) + np.pi / azi_n
    Bm, Am = np.meshgrid(beta, alpha, indexing="ij")
    Bm, Am = Bm.flatten(), Am.flatten()
    s2 = np.stack([np.sin(Bm) * np.cos(Am), np.sin(Bm) * np.sin(Am), np.cos(Bm)], axis=1)
    scale = np.reshape(np.arange(rad_n) / rad_n + 1 / (2 * rad_n), [rad_n, 1, 1])
    return torch.FloatTensor(scale * s2[None]).view(-1, 3)


def bn_eval(x, sd, prefix, affine=True):
    w = sd[prefix + ".weight"] if affine else None
    b = sd[prefix + ".bias"] if affine else None
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], w, b, False, 0.0, 1e-5)


def pad_cyl(x):
    """pad_image / pad_image_3d with kernel 3 (common.py:230-275): circular +-1 on the last (azimuth) axis, zeros +-1 on elevation"""
    x = torch.cat([x[..., -1:], x, x[..., :1]], dim=-1)
    z = torch.zeros_like(x[..., :1, :])
    return torch.cat([z, x, z], dim=-2)


def forward(sd, pts, kpts, des_r, perm):
    """pts (N,3), kpts (K,3), perm (N,) -> desc (K,32); intermediates in a dict for stage-wise checks"""
    p = pts[perm]                                                               # patch_embedder.py:99-100
    idx, nn = ball_query_first_k(kpts, p, 512, des_r)                           # :104-110
    invalid = (idx == -1).float()[..., None]
    patch = nn * (1 - invalid) + kpts[:, None, :] * invalid                     # :122-131
    center = patch[:, -1, :]                                                    # :142
    delta = (patch - center[:, None, :]) / des_r                                # :143, :185-188
    vox = voxel_centres()
    Kn = kpts.shape[0]
    samples = torch.zeros(Kn, 420, 10, 3)
    for k in range(Kn):                                                         # sphere_query, common.py:396-440
        gi, sn = ball_query_first_k(vox, delta[k], 10, 0.8 / 3)
        mask = (gi == gi[:, :1]).float(); mask[:, 0] = 0
        mask[:, 0] += (gi[:, 0] == 0).float()
        samples[k] = sn * (1 - mask[..., None])
    ang = -torch.arange(20, dtype=torch.float64) * (2 * np.pi / 20)             # var_to_invar, common.py:443-469
    R = torch.zeros(20, 3, 3, dtype=torch.float64)
    R[:, 0, 0] = torch.cos(ang); R[:, 0, 1] = -torch.sin(ang); R[:, 1, 0] = torch.sin(ang); R[:, 1, 1] = torch.cos(ang); R[:, 2, 2] = 1
    R = R.float()
    s = samples.view(Kn, 3, 7, 20, 10, 3)
    inv = torch.matmul(s, R.transpose(-1, -2)[None, None, None]).view(Kn, 420, 10, 3)
    x = inv.permute(0, 3, 1, 2)                                                 # (K,3,420,10)  patch_embedder.py:74
    x = F.relu(bn_eval(F.conv2d(x, sd["pnt_layer.0.weight"], sd["pnt_layer.0.bias"]), sd, "pnt_layer.1"))
    x = x.max(dim=3).values.view(Kn, 16, 3, 7, 20)                              # :76-78
    x0 = x
    x = F.conv3d(pad_cyl(x), sd["conv_net.ops.0.weight"], sd["conv_net.ops.0.bias"])          # patchnet.py:52-61
    x = F.relu(bn_eval(x, sd, "conv_net.ops.1", affine=False)).squeeze(2)
    for i in range(1, 8):
        op = 3 * i
        x = F.conv2d(pad_cyl(x), sd[f"conv_net.ops.{op}.weight"], sd[f"conv_net.ops.{op}.bias"])
        if i < 7:
            x = F.relu(bn_eval(x, sd, f"conv_net.ops.{op + 1}", affine=False))
    w = F.relu(bn_eval(F.conv2d(x, sd["pool_layer.0.weight"], sd["pool_layer.0.bias"]), sd, "pool_layer.1"))
    w = F.relu(bn_eval(F.conv2d(w, sd["pool_layer.3.weight"], sd["pool_layer.3.bias"]), sd, "pool_layer.4"))
    f = (x * w).mean(dim=(2, 3))                                                # patch_embedder.py:82
    desc = F.normalize(f, p=2, dim=1)
    return {"desc": desc, "patches": delta, "x0": x0, "equi_raw": x}
