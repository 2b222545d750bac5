"""Deterministic synthetic inputs and random-init weights for the sampling hot path.

There are no datasets or checkpoints offline, so tests and ``bench.py`` use
seeded synthetic scan pairs whose value ranges mirror what the reference's
dataset transform produces (reference ``rectified_point_flow/data/dataset.py:733-900``:
per-part centring, anchor part keeps its pose, whole sample scaled so the anchor
fits the unit cube with a 1.5 margin) and weights drawn per tensor *by name*
(so the same tensors can be loaded into the reference's ``PointCloudDiT`` with
``load_state_dict`` -- see ``oracle/make_golden.py``).

Everything here runs on the CPU generator (mt19937) so that the CPU oracle and
the GPU path see bit-identical inputs; ``x_1`` is always passed explicitly
(reference ``modeling.py:664`` would draw it with the device generator).
"""
from __future__ import annotations

import hashlib
import math
from typing import Sequence

import torch

RAP_12 = dict(embed_dim=512, num_layers=12, num_heads=8, local_feat_dim=32)   # config/model/flow_model/point_cloud_dit_12.yaml
RAP_10 = dict(embed_dim=512, num_layers=10, num_heads=8, local_feat_dim=32)
RAP_16 = dict(embed_dim=512, num_layers=16, num_heads=8, local_feat_dim=32)


def embed_in_dim(cfg) -> int:
    """63 (cond PE) + 63 (x_t PE) + in_dim (latent point features, 0 in every shipped config) + 21 (scale PE, if scale_emb_on)
    + local_feat_dim (if local_feat_concat_on)  (embedding.py:107-118)."""
    return (63 + 63 + int(cfg.get("in_dim", 0)) + (21 if cfg.get("scale_emb_on", True) else 0)
            + (cfg["local_feat_dim"] if cfg.get("local_feat_concat_on", True) else 0))


def weight_spec(cfg) -> list[tuple[str, tuple[int, ...]]]:
    """(name, shape) of every tensor in ``PointCloudDiT.state_dict()`` in the reference's order
    (point_cloud_dit.py:83-117, layer.py:71-89, norm.py:47-58)."""
    d, L, H = cfg["embed_dim"], cfg["num_layers"], cfg["num_heads"]
    dh = d // H
    spec = [("anchor_part_emb.weight", (2, d)),
            ("encoding_manager.emb_proj.weight", (d, embed_in_dim(cfg))),
            ("encoding_manager.emb_proj.bias", (d,))]
    for i in range(L):
        p = f"transformer_layers.{i}."
        for a in ("self", "global"):
            spec += [(p + f"{a}_prenorm.timestep_embedder.linear_1.weight", (d, 256)),
                     (p + f"{a}_prenorm.timestep_embedder.linear_1.bias", (d,)),
                     (p + f"{a}_prenorm.timestep_embedder.linear_2.weight", (d, d)),
                     (p + f"{a}_prenorm.timestep_embedder.linear_2.bias", (d,)),
                     (p + f"{a}_prenorm.linear.weight", (2 * d, d)),
                     (p + f"{a}_prenorm.linear.bias", (2 * d,)),
                     (p + f"{a}_qkv_proj.weight", (3 * d, d)),
                     (p + f"{a}_out_proj.weight", (d, d)),
                     (p + f"{a}_out_proj.bias", (d,))]
            if cfg.get("qk_norm", True):                       # layer.py:75-77, 83-85: the norms exist only with qk_norm=True
                spec += [(p + f"{a}_q_norm.gamma", (H, dh)),
                         (p + f"{a}_k_norm.gamma", (H, dh))]
        spec += [(p + "ff_norm.weight", (d,)), (p + "ff_norm.bias", (d,)),
                 (p + "ff.net.0.proj.weight", (8 * d, d)), (p + "ff.net.0.proj.bias", (8 * d,)),
                 (p + "ff.net.2.weight", (d, 4 * d)), (p + "ff.net.2.bias", (d,))]
    spec += [("final_mlp.0.weight", (d, d)), ("final_mlp.0.bias", (d,)),
             ("final_mlp.2.weight", (d // 2, d)), ("final_mlp.2.bias", (d // 2,)),
             ("final_mlp.4.weight", (3, d // 2))]
    return spec


def _name_seed(name: str, seed: int) -> int:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:7], "little")


def make_weights(cfg, seed: int = 0) -> dict[str, torch.Tensor]:
    """Random-init fp32 weights of the reference architecture, one CPU generator per tensor name.

    Linear weights/biases ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (the nn.Linear default range);
    qk-norm gammas and LayerNorm gains ~ U(0.5, 1.5) (so those code paths are exercised, the
    reference initialises them to 1); LayerNorm bias ~ U(-0.1, 0.1); anchor embedding ~ N(0, 1).
    """
    sd = {}
    for name, shape in weight_spec(cfg):
        g = torch.Generator().manual_seed(_name_seed(name, seed))
        if name == "anchor_part_emb.weight":
            w = torch.randn(shape, generator=g)
        elif name.endswith("gamma") or name.endswith("ff_norm.weight"):
            w = torch.rand(shape, generator=g) + 0.5
        elif name.endswith("ff_norm.bias"):
            w = (torch.rand(shape, generator=g) - 0.5) * 0.2
        else:
            fan_in = shape[1] if len(shape) == 2 else _bias_fan_in(name, cfg)
            bound = 1.0 / math.sqrt(fan_in)
            w = (torch.rand(shape, generator=g) * 2 - 1) * bound
        sd[name] = w.to(torch.float32).contiguous()
    return sd


ENVELOPE_KINDS = ("big_geglu", "tiny_geglu", "mixed_gains")


def envelope_weights(cfg, seed: int, kind: str) -> dict[str, torch.Tensor]:
    """Seeded weights pushed to the edges of the split-precision / bounded-softmax envelope -- the SAME function of
    its inputs up to rounding in the first two cases, so the reference's fp32 result stays O(1) while an intermediate leaves fp16's comfort zone:
      big_geglu    ff.net.0.proj (weight, bias) x 48 and ff.net.2.weight / 48^2: the GEGLU output h * gelu(g) grows ~2 300 x (10^3 ... 10^4,
                   toward fp16's 65 504) and the down-projection takes it back;
      tiny_geglu   ff.net.0.proj / 64 and ff.net.2.weight x 64^2: GEGLU outputs of 10^-5 ... 10^-3, whose fp16 TAILS (x 2^-11) are subnormal or
                   zero -- the values themselves straddle fp16's smallest normal number 6.1e-5;
      mixed_gains  the q / k gains of layer 0's per-part attention and layer 1's per-sample attention x 2.5 (logit bound 8 max|gq| max|gk| up
                   to ~110 > 40: online softmax), the other two launches untouched (bound <= 18: bounded softmax) -- kernel choice per launch."""
    sd = make_weights(cfg, seed)
    L = cfg["num_layers"]
    if kind == "big_geglu" or kind == "tiny_geglu":
        a = 48.0 if kind == "big_geglu" else 1.0 / 64.0
        for i in range(L):
            p = f"transformer_layers.{i}."
            sd[p + "ff.net.0.proj.weight"] = sd[p + "ff.net.0.proj.weight"] * a
            sd[p + "ff.net.0.proj.bias"] = sd[p + "ff.net.0.proj.bias"] * a
            sd[p + "ff.net.2.weight"] = sd[p + "ff.net.2.weight"] / (a * a)
    elif kind == "mixed_gains":
        for i, which in ((0, "self"), (1 % L, "global")):
            p = f"transformer_layers.{i}.{which}_"
            sd[p + "q_norm.gamma"] = sd[p + "q_norm.gamma"] * 2.5
            sd[p + "k_norm.gamma"] = sd[p + "k_norm.gamma"] * 2.5
    else:
        raise ValueError(f"unknown envelope kind {kind!r} (one of {ENVELOPE_KINDS})")
    return sd


def _bias_fan_in(name: str, cfg) -> int:
    d = cfg["embed_dim"]
    if "emb_proj" in name:
        return embed_in_dim(cfg)
    if "linear_1" in name:
        return 256
    if "ff.net.2" in name:
        return 4 * d
    if name == "final_mlp.2.bias":
        return d
    return d


def _random_rotation(g: torch.Generator) -> torch.Tensor:
    q = torch.randn(4, generator=g, dtype=torch.float64)
    q = q / q.norm()
    w, x, y, z = q.tolist()
    return torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=torch.float64)


def _box_surface(n: int, g: torch.Generator) -> torch.Tensor:
    """n points on the surfaces of 3 random boxes inside [-0.5, 0.5]^3."""
    pts = []
    per = [n // 3, n // 3, n - 2 * (n // 3)]
    for m in per:
        half = torch.rand(3, generator=g, dtype=torch.float64) * 0.2 + 0.05
        ctr = (torch.rand(3, generator=g, dtype=torch.float64) - 0.5) * (1 - 2 * half)
        u = torch.rand(m, 3, generator=g, dtype=torch.float64) * 2 - 1
        face = torch.randint(0, 3, (m,), generator=g)
        sign = torch.randint(0, 2, (m,), generator=g).to(torch.float64) * 2 - 1
        u[torch.arange(m), face] = sign
        pts.append(ctr + u * half)
    return torch.cat(pts, 0)


def make_sample(part_sizes: Sequence[int], g: torch.Generator):
    """One multi-view sample: returns (cond (n,3), gt (n,3)) float64, parts concatenated."""
    n_tot = int(sum(part_sizes))
    scene = _box_surface(max(3 * max(part_sizes), 64), g)
    conds, gts = [], []
    for p, n in enumerate(part_sizes):
        if n == 0:
            continue
        nrm = torch.randn(3, generator=g, dtype=torch.float64)
        nrm = nrm / nrm.norm()
        proj = scene @ nrm
        thr = torch.quantile(proj, 0.4)
        vis = torch.nonzero(proj >= thr).squeeze(1)
        idx = vis[torch.randint(0, vis.numel(), (n,), generator=g)]
        gt = scene[idx]
        if p == 0:
            cond = gt.clone()                                   # anchor keeps its pose (dataset.py:867)
        else:
            cond = (gt - gt.mean(0, keepdim=True)) @ _random_rotation(g).T   # centred per part (dataset.py:804)
        conds.append(cond)
        gts.append(gt)
    cond = torch.cat(conds, 0)
    gt = torch.cat(gts, 0)
    n0 = part_sizes[0] if part_sizes[0] > 0 else n_tot
    s = 1.0 / (cond[:n0].abs().max() * 1.5)                     # dataset.py:780,791
    return cond * s, gt * s


def make_inputs(parts: Sequence[Sequence[int]], seed: int = 1234, feat_dim: int = 32,
                max_parts: int | None = None) -> dict[str, torch.Tensor]:
    """Packed batch in the reference's collate schema (data/datamodule.py:169-198):

    pointclouds (TP,3) f32, pointclouds_gt (TP,3) f32, features (TP,feat_dim) f32 unit-norm,
    scales (B,) f32, anchor_indices (TP,) bool, points_per_part (B,P) int64, cu_seqlens (B+1,) int64,
    plus x_1 (TP,3) f32 -- the explicit initial noise.
    """
    B = len(parts)
    P = max_parts if max_parts is not None else max(len(p) for p in parts)
    ppp = torch.zeros(B, P, dtype=torch.int64)
    conds, gts, feats, anchors, x1s, lens = [], [], [], [], [], []
    scales = torch.empty(B, dtype=torch.float32)
    for b, sizes in enumerate(parts):
        g = torch.Generator().manual_seed(seed + b)
        for p, n in enumerate(sizes):
            ppp[b, p] = n
        cond, gt = make_sample(list(sizes), g)
        n = cond.shape[0]
        f = torch.randn(n, feat_dim, generator=g, dtype=torch.float64)
        f = f / f.norm(dim=1, keepdim=True)                     # MiniSpinNet output is unit-norm (patch_embedder.py:83)
        a = torch.zeros(n, dtype=torch.bool)
        a[: sizes[0]] = True
        scales[b] = float(torch.rand(1, generator=g).item() * 45 + 5)
        x1 = torch.randn(n, 3, generator=g, dtype=torch.float32)
        conds.append(cond.float()); gts.append(gt.float()); feats.append(f.float()); anchors.append(a)
        x1s.append(x1); lens.append(n)
    cu = torch.zeros(B + 1, dtype=torch.int64)
    cu[1:] = torch.cumsum(torch.tensor(lens, dtype=torch.int64), 0)
    return {"pointclouds": torch.cat(conds).contiguous(), "pointclouds_gt": torch.cat(gts).contiguous(),
            "features": torch.cat(feats).contiguous(), "scales": scales, "anchor_indices": torch.cat(anchors),
            "points_per_part": ppp, "cu_seqlens": cu, "x_1": torch.cat(x1s).contiguous()}


def make_inputs_subset(parts: Sequence[Sequence[int]], indices: Sequence[int], seed: int = 1234, feat_dim: int = 32,
                       max_parts: int | None = None) -> dict[str, torch.Tensor]:
    """The samples ``indices`` of ``make_inputs(parts, seed)`` as one packed batch, WITHOUT generating the others (sample b is drawn from
    ``seed + b``, so a rank of a sharded job builds exactly its own samples of the global batch).  ``max_parts`` defaults to the largest
    part count of the WHOLE job, so every rank pads its points_per_part table to the same width."""
    P = max_parts if max_parts is not None else max(len(p) for p in parts)
    one = [make_inputs([parts[i]], seed=seed + i, feat_dim=feat_dim, max_parts=P) for i in indices]
    if not one:
        raise ValueError("empty shard")
    out = {k: torch.cat([o[k] for o in one]) for k in one[0] if k != "cu_seqlens"}
    cu = torch.zeros(len(one) + 1, dtype=torch.int64)
    cu[1:] = torch.cumsum(torch.tensor([int(o["cu_seqlens"][-1]) for o in one], dtype=torch.int64), 0)
    out["cu_seqlens"] = cu
    return out


def make_uniform_inputs(batch: int, views: int, n_points: int, seed: int = 1234, feat_dim: int = 32):
    """BASELINE.json geometry: ``batch`` samples of ``views`` x ``n_points``."""
    return make_inputs([[n_points] * views for _ in range(batch)], seed=seed, feat_dim=feat_dim)


def ragged_regime_parts(total_points: int = 262144, seed: int = 4321) -> list[list[int]]:
    """Part sizes of a RAGGED packed batch in the reference's own regime (config/RAP_inference.yaml:30-36, demo.py:568-571: 2 ... 512
    parts per sample, 200 ... 20 000 points per part after voxel-adaptive FPS, <= 400 000 points per batch): samples with P = 2, 8
    and 64 parts in turn, part sizes log-uniform in a range per kind (two scans: 2 000 ... 20 000; eight views: 500 ... 8 000;
    sixty-four fragments: 200 ... 1 200), until ``total_points`` is reached -- the last part is cut so that the total is
    ``total_points - 133`` (NOT a multiple of any tile size).  Seeded (CPU generator); about 262 k points by default = the token
    count of BASELINE configs[1]."""
    g = torch.Generator().manual_seed(seed)
    kinds = [(2, 2000, 20000), (8, 500, 8000), (64, 200, 1200)]
    target = int(total_points) - 133
    parts, total, k = [], 0, 0
    while total < target:
        P, lo, hi = kinds[k % len(kinds)]
        k += 1
        u = torch.rand(P, generator=g, dtype=torch.float64)
        sizes = [int(round(math.exp(math.log(lo) + float(x) * (math.log(hi) - math.log(lo))))) for x in u]
        room = target - total
        if sum(sizes) > room:                       # cut the last sample to the budget (drop parts that no longer fit)
            cut, acc = [], 0
            for n in sizes:
                if acc + n <= room:
                    cut.append(n); acc += n
                elif room - acc >= 200:
                    cut.append(room - acc); acc = room
                    break
            if not cut:
                if parts:
                    parts[-1][-1] += room           # a sliver: give it to the previous sample's last part
                    total += room
                break
            sizes = cut
        parts.append(sizes)
        total += sum(sizes)
    return parts
