"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.npz by running the reference's OWN modules.

Run inside the build container (needs /root/reference):   python -m oracle.make_golden
The fixtures hold the inputs, the seeds/config that regenerate the weights (rap_amd.synthetic.make_weights)
and the outputs of the unmodified reference code (sampler.py, flow_model/*, procrustes.py imported by
oracle/ref_loader.py; fp32, CPU).  They travel to the GPU box, where /root/reference does not exist.
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_loader                      # noqa: E402
from oracle import rap_oracle as O                 # noqa: E402
from rap_amd import synthetic as S                 # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = [
    # name, num_layers, parts, input seed, weight seed, steps, rigidity
    ("l2_ragged_rigid", 2, [[37, 64, 100], [50, 129]], 7, 0, 4, True),
    ("l2_ragged_free", 2, [[37, 64, 100], [50, 129]], 7, 0, 4, False),
    ("l2_emptypart_rigid", 2, [[70, 45, 0], [33, 90, 61]], 11, 1, 3, True),
    ("l12_small_rigid", 12, [[64, 96], [128, 40, 33]], 21, 0, 3, True),
    ("l12_pair512_free", 12, [[512, 512]], 31, 0, 2, False),
    # the reference's other two model sizes (config/model/flow_model/point_cloud_dit_{10,16}.yaml): rap_10 and rap_16
    ("l16_small_rigid", 16, [[96, 64], [40, 128, 33]], 41, 2, 3, True),
    ("l10_small_free", 10, [[80, 80], [150, 30]], 43, 3, 3, False),
    # round 5: the constructor switches of PointCloudDiT every shipped config leaves True (point_cloud_dit.py:28,33-34) -- one fixture each
    ("l2_noqknorm_rigid", 2, [[60, 41], [130, 20, 77]], 51, 4, 3, True),
    ("l2_noscale_free", 2, [[37, 64, 100], [50, 129]], 53, 5, 3, False),
    ("l2_nofeat_rigid", 2, [[90, 33], [45, 45, 45]], 55, 6, 3, True),
    # round 6: in_dim = 64 -- latent point features concatenated into the embedding (embedding.py:107-118,163-166; modeling.py:788)
    ("l2_latent64_rigid", 2, [[70, 51], [40, 120, 33]], 57, 7, 3, True),
]
# name -> PointCloudDiT keyword overrides
CASE_SWITCHES = {"l2_noqknorm_rigid": {"qk_norm": False}, "l2_noscale_free": {"scale_emb_on": False},
                 "l2_nofeat_rigid": {"local_feat_concat_on": False}, "l2_latent64_rigid": {"in_dim": 64}}


def weights_checksum(sd) -> float:
    return float(sum(v.double().sum().item() for v in sd.values()))


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--case=")]
    for name, L, parts, iseed, wseed, steps, rigid in CASES:
        if only and name not in only:
            continue
        cfg = dict(S.RAP_12); cfg["num_layers"] = L
        cfg.update(CASE_SWITCHES.get(name, {}))
        sd = S.make_weights(cfg, wseed)
        inp = S.make_inputs(parts, seed=iseed)
        if cfg.get("in_dim", 0):      # latent point features as a PTv3 encoder would hand them over: (TP, in_dim), O(1) values
            inp["latent_features"] = torch.randn(inp["x_1"].shape[0], cfg["in_dim"], generator=torch.Generator().manual_seed(1000 + iseed))
        ref = ref_loader.reference_sample(cfg, sd, inp, steps, rigid)
        # one stand-alone forward of the reference PointCloudDiT with a different t per sample
        model = ref_loader.build_reference_dit(cfg, sd)
        cu_b, cu_p = O.prepare_cu_seqlens(inp)
        B = len(parts)
        ts = torch.linspace(0.15, 0.9, B)
        with torch.inference_mode():
            fw = model(x=inp["x_1"], timesteps=ts, cond_coord=inp["pointclouds"], local_features=inp["features"],
                       latent_features=inp.get("latent_features"), scales=inp["scales"], anchor_indices=inp["anchor_indices"],
                       cu_seqlens_batch=cu_b, cu_seqlens_part=cu_p, return_transformer_features=True)
        out = {
            "num_layers": np.int64(L), "weight_seed": np.int64(wseed), "num_steps": np.int64(steps),
            "rigidity": np.int64(int(rigid)), "weights_checksum": np.float64(weights_checksum(sd)),
            "fwd_timesteps": ts.numpy(), "fwd_velocity": fw["velocity"].numpy(),
            "fwd_features": fw["transformer_features"].numpy(),
            "end_point_trajectory": ref["end_point_trajectory"].numpy(), "trajectory": ref["trajectory"].numpy(),
            "R": ref["R"].numpy(), "t": ref["t"].numpy(),
            # transformer_features captured by the sampling call itself (modeling.py:678-695: model call steps-1, t = dt)
            "sample_features": ref["transformer_features"].numpy(), "sample_features_timestep": np.float64(ref["features_timestep"]),
        }
        for k, v in inp.items():
            out["in_" + k] = v.numpy()
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path) // 1024, "KiB")


def make_selection_golden():
    """Generation selection by rigidity: 3 generations of the l2_ragged_rigid case with
    rigidity forcing OFF (so the trajectories are not rigid and the RMSEs differ), a sample with an empty part, averaged
    over the trajectory steps and at the final step; reference = its own fit_transformations + compute_rigidity_rmse."""
    cfg = dict(S.RAP_12); cfg["num_layers"] = 2
    sd = S.make_weights(cfg, 1)
    inp = S.make_inputs([[70, 45, 0], [33, 90, 61], [128, 40]], seed=13)
    cu_b, _ = O.prepare_cu_seqlens(inp)
    trajs = []
    for seed in (104, 109, 101):      # chosen so that the three objects pick three different generations
        gen = torch.Generator().manual_seed(seed)
        x_1 = torch.randn(inp["x_1"].shape, generator=gen)
        inp_g = dict(inp); inp_g["x_1"] = x_1
        trajs.append(ref_loader.reference_sample(cfg, sd, inp_g, 4, False)["end_point_trajectory"])
    out = {"num_layers": np.int64(2), "weight_seed": np.int64(1), "num_steps": np.int64(4)}
    for k, v in inp.items():
        out["in_" + k] = v.numpy()
    out["trajectories"] = torch.stack(trajs).numpy()
    for tag, use_avg in (("avg", True), ("final", False)):
        r = ref_loader.reference_generation_selection(inp["pointclouds"], trajs, inp["points_per_part"], cu_b.long(), inp["scales"],
                                                      use_average=use_avg)
        for k, v in r.items():
            out[f"{tag}_{k}"] = v.numpy()
    ref = ref_loader.load_reference()
    Rf, tf = ref.fit_transformations(inp["pointclouds"], trajs[0][-1], inp["points_per_part"], cu_b.long())
    out["per_part_rmse"] = ref.compute_rigidity_rmse(inp["pointclouds"], trajs[0][-1], Rf, tf, inp["points_per_part"], cu_b.long(),
                                                     None, average_per_part=True).numpy()
    path = os.path.join(GOLDEN_DIR, "selection_g3.npz")
    np.savez_compressed(path, **out)
    print("selection_g3", os.path.getsize(path) // 1024, "KiB")


def make_overlap_golden():
    """Cross-part overlap ratio: the reference's own compute_overlap_ratio on 4 objects (two
    overlapping views of one surface, a single-part object -> 0, a trailing empty part, a 1-point part), thresholds chosen
    inside the distance distribution so that the ratios are neither 0 nor 1."""
    g = torch.Generator().manual_seed(5)
    ppp = torch.tensor([[400, 300, 0], [500, 0, 0], [257, 1, 130], [64, 64, 64]])
    clouds = []
    for b in range(ppp.shape[0]):
        surf = torch.rand(2000, 3, generator=g); surf[:, 2] = 0.2 * torch.sin(3 * surf[:, 0])       # one shared surface per object
        for p in range(ppp.shape[1]):
            idx = torch.randint(0, 2000, (int(ppp[b, p]),), generator=g)
            clouds.append(surf[idx] + 0.002 * torch.randn(int(ppp[b, p]), 3, generator=g))
    pred = torch.cat(clouds)
    cu = torch.cat([torch.zeros(1, dtype=torch.long), ppp.sum(1).cumsum(0)])
    taus = [0.005, 0.01, 0.02]
    ref = ref_loader.load_reference()
    ratios = ref.compute_overlap_ratio(pred, ppp, cu, taus)
    path = os.path.join(GOLDEN_DIR, "overlap_ratio.npz")
    np.savez_compressed(path, pred=pred.numpy(), points_per_part=ppp.numpy(), cu_seqlens=cu.numpy(), taus=np.array(taus),
                        ratios=ratios.numpy())
    print("overlap_ratio", os.path.getsize(path) // 1024, "KiB", ratios)


def make_envelope_goldens():
    """the envelope of the split-precision mode and of the per-launch softmax selection against the REFERENCE, not just the
    kernels.  Three L = 2 fixtures of the unmodified reference (fp32, CPU) on weights from rap_amd.synthetic.envelope_weights; the fixture also
    records how large the GEGLU output actually gets (read off the oracle's evaluation of the same weights)."""
    cfg = dict(S.RAP_12); cfg["num_layers"] = 2
    parts, iseed, wseed, steps = [[300, 211], [64, 500, 33]], 61, 8, 3
    inp = S.make_inputs(parts, seed=iseed)
    cu_b, cu_p = O.prepare_cu_seqlens(inp)
    for kind in S.ENVELOPE_KINDS:
        sd = S.envelope_weights(cfg, wseed, kind)
        ref = ref_loader.reference_sample(cfg, sd, inp, steps, True)
        model = ref_loader.build_reference_dit(cfg, sd)
        ts = torch.linspace(0.15, 0.9, len(parts))
        with torch.inference_mode():
            fw = model(x=inp["x_1"], timesteps=ts, cond_coord=inp["pointclouds"], local_features=inp["features"], latent_features=None,
                       scales=inp["scales"], anchor_indices=inp["anchor_indices"], cu_seqlens_batch=cu_b, cu_seqlens_part=cu_p)
        # magnitude of the GEGLU output of layer 0 at the forward's inputs (oracle arithmetic; information only)
        taps = {}
        O.dit_forward(sd, cfg, inp["x_1"], ts, inp["pointclouds"], inp["features"], inp["scales"], inp["anchor_indices"], cu_b, cu_p, taps=taps)
        p0 = "transformer_layers.0."
        x = torch.nn.functional.layer_norm(taps["l0_after_global_attn"], (512,), sd[p0 + "ff_norm.weight"], sd[p0 + "ff_norm.bias"], eps=1e-5)
        u = torch.nn.functional.linear(x, sd[p0 + "ff.net.0.proj.weight"], sd[p0 + "ff.net.0.proj.bias"])
        hh, gg = u.chunk(2, dim=-1)
        ge = (hh * torch.nn.functional.gelu(gg)).abs()
        out = {"num_layers": np.int64(2), "weight_seed": np.int64(wseed), "num_steps": np.int64(steps), "weights_checksum": np.float64(weights_checksum(sd)),
               "fwd_timesteps": ts.numpy(), "fwd_velocity": fw.numpy(), "end_point_trajectory": ref["end_point_trajectory"].numpy(),
               "trajectory": ref["trajectory"].numpy(), "R": ref["R"].numpy(), "t": ref["t"].numpy(),
               "geglu_abs_max": np.float64(ge.max()), "geglu_abs_median": np.float64(ge.median()), "geglu_abs_p01": np.float64(ge.flatten().kthvalue(max(1, ge.numel() // 100)).values)}
        for k, v in inp.items():
            out["in_" + k] = v.numpy()
        path = os.path.join(GOLDEN_DIR, f"envelope_{kind}.npz")
        np.savez_compressed(path, **out)
        print(f"envelope_{kind}", os.path.getsize(path) // 1024, "KiB; |GEGLU| max", float(ge.max()), "median", float(ge.median()), "; max|v|", float(fw.abs().max()))


def make_transform_errors_golden():
    """compute_transform_errors, no-ICP branch (eval/metrics.py:165-303): the reference's OWN function on 6
    samples x 5 parts -- random proper rotations with errors from a fraction of a degree to ~170 degrees, a sample whose anchor is not
    part 0, trailing empty parts, a sample with two anchors (the first counts), one WITHOUT an anchor (identity frame), one with only
    the anchor (0 / 0 = NaN), with and without matched_part_ids and scale."""
    g = torch.Generator().manual_seed(77)
    B, P = 6, 5

    def rot(angle_deg=None):
        q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
        if torch.det(q) < 0:
            q[:, 0] *= -1
        if angle_deg is None:
            return q
        axis = torch.randn(3, generator=g, dtype=torch.float64); axis /= axis.norm()
        K = torch.tensor([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]], dtype=torch.float64)
        a = math.radians(angle_deg)
        return torch.eye(3, dtype=torch.float64) + math.sin(a) * K + (1 - math.cos(a)) * (K @ K)
    ppp = torch.tensor([[400, 300, 50, 0, 0], [500, 120, 0, 0, 0], [257, 1, 130, 64, 9], [64, 64, 64, 0, 0], [100, 0, 0, 0, 0], [30, 40, 50, 60, 0]])
    anchor = torch.zeros(B, P, dtype=torch.bool)
    anchor[0, 0] = True; anchor[1, 1] = True; anchor[2, 2] = True; anchor[2, 4] = True; anchor[4, 0] = True      # sample 3 and 5: no anchor at all (5) / none (3)
    anchor[5, 3] = True
    R_gt = torch.stack([torch.stack([rot() for _ in range(P)]) for _ in range(B)]).float()
    t_gt = torch.randn(B, P, 3, generator=g)
    angles = [0.3, 2.0, 11.0, 45.0, 90.0, 170.0]
    R_pred = torch.stack([torch.stack([(rot(angles[(b + p) % len(angles)]) @ R_gt[b, p].double()) for p in range(P)]) for b in range(B)]).float()
    t_pred = t_gt + 0.05 * torch.randn(B, P, 3, generator=g)
    scale = torch.rand(B, generator=g) * 3 + 0.5
    matched = torch.stack([torch.randperm(P, generator=g) for _ in range(B)])
    cu = torch.cat([torch.zeros(1, dtype=torch.long), ppp.sum(1).cumsum(0)])
    pts = torch.randn(int(cu[-1]), 3, generator=g)
    ref = ref_loader.load_reference()
    out = {"R_gt": R_gt.numpy(), "t_gt": t_gt.numpy(), "R_pred": R_pred.numpy(), "t_pred": t_pred.numpy(), "points_per_part": ppp.numpy(),
           "anchor_part": anchor.numpy(), "scale": scale.numpy(), "matched_part_ids": matched.numpy(), "cu_seqlens": cu.numpy()}
    for tag, mid, sc in (("plain", None, None), ("scaled", None, scale), ("matched", matched, scale)):
        re, te = ref.metrics.compute_transform_errors(pts, pts, R_gt, t_gt, R_pred, t_pred, ppp, anchor, matched_part_ids=mid, scale=sc,
                                                      cu_seqlens_batch=cu, use_icp=False)
        out[f"{tag}_rot"] = re.numpy(); out[f"{tag}_trans"] = te.numpy()
    path = os.path.join(GOLDEN_DIR, "transform_errors.npz")
    np.savez_compressed(path, **out)
    print("transform_errors", os.path.getsize(path) // 1024, "KiB", out["plain_rot"], out["plain_trans"])


def make_spinnet_golden():
    """MiniSpinNet descriptors: the reference's unmodified module (ball_query restated, see
    ref_loader.reference_spinnet_forward) on a 3000-point surface, 16 keypoints whose balls hold both fewer and more than 512
    points (so both the keypoint fill and the 512-cap / last-slot centring are exercised), seeded synthetic weights."""
    from rap_amd.spinnet import make_spinnet_weights
    from oracle import spinnet_oracle as SO
    sd = make_spinnet_weights(0)
    g = torch.Generator().manual_seed(1)
    pts = torch.rand(3000, 3, generator=g)
    pts[:, 2] = 0.3 * torch.sin(4 * pts[:, 0]) + 0.05 * torch.rand(3000, generator=g)
    kpts = pts[torch.randperm(3000, generator=g)[:16]].clone()
    kpts[0] = torch.tensor([5.0, 5.0, 5.0])                 # a keypoint with an empty ball: the patch is 512 copies of it
    des_r, seed = 0.3, 5
    ref = ref_loader.reference_spinnet_forward(sd, pts, kpts, des_r, seed)
    counts = (((kpts[:, None] - pts[None]) ** 2).sum(-1) < des_r ** 2).sum(1)
    path = os.path.join(GOLDEN_DIR, "spinnet_k16.npz")
    np.savez_compressed(path, pts=pts.numpy(), kpts=kpts.numpy(), perm=ref["perm"], des_r=np.float64(des_r), weight_seed=np.int64(0),
                        desc=ref["desc"].numpy(), patches_first=ref["patches"][:, :4].numpy(), patches_last=ref["patches"][:, -1].numpy(),
                        ball_counts=counts.numpy())
    print("spinnet_k16", os.path.getsize(path) // 1024, "KiB; points per ball:", counts.tolist())
    # the same cloud with every patch aligned to its own normal (is_aligned_to_global_z = False: cal_Z_axis + RodsRotatFormula)
    ref2 = ref_loader.reference_spinnet_forward(sd, pts, kpts, des_r, seed, is_aligned_to_global_z=False)
    path2 = os.path.join(GOLDEN_DIR, "spinnet_k16_lrf.npz")
    np.savez_compressed(path2, pts=pts.numpy(), kpts=kpts.numpy(), perm=ref2["perm"], des_r=np.float64(des_r), weight_seed=np.int64(0),
                        desc=ref2["desc"].numpy(), R=ref2["R"].numpy(), patches_first=ref2["patches"][:, :4].numpy())
    print("spinnet_k16_lrf", os.path.getsize(path2) // 1024, "KiB")


def make_nn_metrics_golden():
    """Nearest-neighbour metrics: the reference's own compute_correspondence_rmse on a scan pair
    (overlapping views of one surface, a slightly wrong predicted pose); chamfer RMSE from the oracle restatement (pytorch3d's
    chamfer_distance is absent) on a 3-object batch."""
    g = torch.Generator().manual_seed(17)
    surf = torch.rand(6000, 3, generator=g); surf[:, 2] = 0.25 * torch.cos(3 * surf[:, 1])
    sg = surf[torch.randint(0, 6000, (900,), generator=g)] + 0.003 * torch.randn(900, 3, generator=g)
    tg = surf[torch.randint(0, 6000, (1100,), generator=g)] + 0.003 * torch.randn(1100, 3, generator=g)
    sp = sg + torch.tensor([0.01, -0.02, 0.005]); tp = tg + 0.002 * torch.randn(1100, 3, generator=g)
    ref = ref_loader.load_reference()
    out = {"source_gt": sg.numpy(), "target_gt": tg.numpy(), "source_pred": sp.numpy(), "target_pred": tp.numpy()}
    for thr in (0.02, 0.05):
        rmse, n, ratio = ref.compute_correspondence_rmse(sg, tg, sp, tp, distance_threshold=thr)
        out[f"corr_{thr}"] = np.array([float(rmse), n, ratio])
    rmse0, n0, r0 = ref.compute_correspondence_rmse(sg, tg + 10.0, sp, tp, distance_threshold=0.05)
    out["corr_none"] = np.array([float(rmse0), n0, r0])
    cu = torch.tensor([0, 700, 1500, 1501])
    gt = torch.rand(1501, 3, generator=g); pred = gt + 0.01 * torch.randn(1501, 3, generator=g)
    out.update({"cd_gt": gt.numpy(), "cd_pred": pred.numpy(), "cd_cu": cu.numpy(), "cd": O.compute_cd(gt, pred, cu).numpy()})
    path = os.path.join(GOLDEN_DIR, "nn_metrics.npz")
    np.savez_compressed(path, **out)
    print("nn_metrics", os.path.getsize(path) // 1024, "KiB", out["corr_0.02"], out["corr_0.05"], out["corr_none"], out["cd"])


def make_voxel_golden():
    """Voxel down-sampling (preprocessing): the reference's own voxel_down_sample_torch on a
    40k-point scan-like cloud (negative coordinates, anisotropic extent) at two voxel sizes; the fixture keeps the seed, not the
    points (regenerated by the test with the same generator calls)."""
    du = ref_loader.load_reference_dataset_utils()
    g = torch.Generator().manual_seed(23)
    p = (torch.rand(40000, 3, generator=g) - 0.4) * torch.tensor([30.0, 22.0, 4.0])
    out = {"seed": np.int64(23), "n": np.int64(40000), "scale": np.array([30.0, 22.0, 4.0], np.float32)}
    for vs in (0.25, 1.0):
        out[f"idx_{vs}"] = du.voxel_down_sample_torch(p, vs).numpy()
    path = os.path.join(GOLDEN_DIR, "voxel_downsample.npz")
    np.savez_compressed(path, **out)
    print("voxel_downsample", os.path.getsize(path) // 1024, "KiB", {k: v.shape for k, v in out.items() if k.startswith("idx")})


def make_transform_golden():
    """Output transform files: the reference's own Evaluator._save_transformation_files on a
    3-object batch (trailing empty part, random GT poses / scales / global frames), with and without the global frame;
    the fixture keeps the inputs and the 4x4 matrices parsed back from the text files it wrote."""
    import tempfile
    from scipy.spatial.transform import Rotation
    g = torch.Generator().manual_seed(77)
    B, P = 3, 3
    ppp = torch.tensor([[70, 45, 0], [33, 90, 61], [128, 40, 0]])
    def rot(n):
        q = torch.randn(n, 4, generator=g)
        return torch.from_numpy(Rotation.from_quat(q.numpy()).as_matrix()).float()
    data = {"rotations": rot(B * P).reshape(B, P, 3, 3), "translations": torch.randn(B, P, 3, generator=g) * 0.3,
            "scales": torch.rand(B, generator=g) * 45 + 5, "points_per_part": ppp}
    R_pred = rot(B * P).reshape(B, P, 3, 3); t_pred = torch.randn(B, P, 3, generator=g) * 0.3
    G_R = rot(B); G_t = torch.randn(B, 3, generator=g) * 20
    out = {"in_" + k: v.numpy() for k, v in data.items()}
    out.update({"R_pred": R_pred.numpy(), "t_pred": t_pred.numpy(), "global_rotation": G_R.numpy(), "global_translation": G_t.numpy(),
                "sample_indices": np.array([7, 8, 12])})
    for tag, gr, gt in (("plain", None, None), ("global", G_R, G_t)):
        with tempfile.TemporaryDirectory() as d:
            files = ref_loader.reference_transformation_files(data, d, "synth", [7, 8, 12], "selected" if tag == "global" else 1,
                                                              R_pred, t_pred, gr, gt)
        out[f"{tag}_names"] = np.array(sorted(files))
        out[f"{tag}_matrices"] = np.stack([files[k] for k in sorted(files)])
    path = os.path.join(GOLDEN_DIR, "transform_files.npz")
    np.savez_compressed(path, **out)
    print("transform_files", os.path.getsize(path) // 1024, "KiB", len(out["plain_names"]), "files per mode")


def collate_samples(seed=3):
    """Raw multi-part samples for the collate fixture: metric-scale scans (tens of metres, large offsets from the origin like UTM
    coordinates), ragged part counts, the largest part not first, a tie for the largest part (argmax takes the first)."""
    g = np.random.RandomState(seed)
    sizes = [[300, 517, 120], [64, 64], [1000, 31, 257, 400]]
    samples = []
    for b, parts in enumerate(sizes):
        centre = g.uniform(-500, 500, size=3) * (b + 1)
        ps, fs = [], []
        for n in parts:
            ps.append(centre + g.uniform(-1, 1, size=3) * 20 + g.normal(size=(n, 3)) * np.array([8.0, 5.0, 1.5]))
            f = g.normal(size=(n, 32)).astype(np.float32)
            fs.append(f / np.linalg.norm(f, axis=1, keepdims=True))
        samples.append({"parts": ps, "features": fs})
    return samples


def make_collate_golden():
    """Input side of the boundary: the reference's own PointCloudDataset._transform (evaluation
    split) + variable_collate_fn on three ragged multi-part samples (ref_loader.reference_transform_and_collate); the fixture keeps
    the raw parts, the numpy seed that reproduces the within-part shuffles, and every arithmetic key of the collated batch."""
    samples = collate_samples()
    max_parts, seed = 5, 11
    ref = ref_loader.reference_transform_and_collate(samples, max_parts, seed)
    out = {"max_parts": np.int64(max_parts), "numpy_seed": np.int64(seed), "num_samples": np.int64(len(samples))}
    for b, smp in enumerate(samples):
        out[f"raw_counts_{b}"] = np.array([len(p) for p in smp["parts"]], dtype=np.int64)
        out[f"raw_points_{b}"] = np.concatenate(smp["parts"])
        out[f"raw_features_{b}"] = np.concatenate(smp["features"])
    for k in ("cu_seqlens", "pointclouds", "pointclouds_gt", "part_indices", "anchor_indices", "features", "rotations", "translations",
              "points_per_part", "anchor_parts", "init_rotation", "scales", "global_rotation", "global_translation"):
        out["ref_" + k] = ref[k].numpy()
    path = os.path.join(GOLDEN_DIR, "collate_transform.npz")
    np.savez_compressed(path, **out)
    print("collate_transform", os.path.getsize(path) // 1024, "KiB", {k: out["ref_" + k].shape for k in ("pointclouds", "translations", "scales")})


def _traj_summary(ref, stride):
    """What a full-geometry fixture keeps of a sampling call: the final registered cloud, the last x_t, the poses, the
    per-step max-norms of both trajectories and every `stride`-th point of every step (so error growth over the re-noised
    steps can be checked step by step without storing S x TP x 3 floats)."""
    ep, tr = ref["end_point_trajectory"], ref["trajectory"]
    extra = {}
    if ref.get("transformer_features") is not None:            # last-call features of the sampling call, every `stride`-th token
        extra = {"sample_features_strided": ref["transformer_features"][::stride].numpy(),
                 "sample_features_max": np.float64(ref["transformer_features"].abs().max()),
                 "sample_features_timestep": np.float64(ref["features_timestep"])}
    return {**extra, "final_end_point": ep[-1].numpy(), "final_x_t": tr[-1].numpy(), "R": ref["R"].numpy(), "t": ref["t"].numpy(),
            "end_point_step_max": ep.abs().amax(dim=(1, 2)).numpy(), "x_t_step_max": tr.abs().amax(dim=(1, 2)).numpy(),
            "stride": np.int64(stride), "end_point_strided": ep[:, ::stride].numpy(), "x_t_strided": tr[:, ::stride].numpy()}


def make_headline_goldens(which=("c1_rigid", "c1_free", "c3", "c4")):
    """Full-geometry, ALL-STEP fixtures at the BASELINE.json geometries, from the reference's unmodified
    modules (ref_loader.reference_sample).  Inputs are `make_uniform_inputs(1, views, points, seed=1234)` = sample 0 of
    bench.py's batch, weights `make_weights(RAP_12, 0)`: the fixtures keep seeds, not inputs.
      headline_c1_rigid / headline_c1_free : configs[1] geometry, 1 pair x 2 x 4096, rap_12, all 20 steps, rigidity on / off
      headline_c3_rigid                    : configs[3] geometry, 1 sample x 8 x 2048, rap_12, all 30 steps, rigidity on
      headline_c4_forward                  : configs[4] geometry, 2 x 32768, ONE forward of a 2-layer model at t = 0.5
    The wall time of the configs[1] call is the CPU baseline of record (live reference, all 20 steps of one pair) and is
    written to profiles/r02_cpu_reference_headline.json."""
    import json, time
    torch.set_num_threads(int(os.environ.get("RAP_GOLDEN_THREADS", os.cpu_count() or 1)))
    timings = {}
    cfg = dict(S.RAP_12)
    sd = S.make_weights(cfg, 0)
    jobs = {"c1_rigid": ("headline_c1_rigid", 2, 4096, 20, True, 32), "c1_free": ("headline_c1_free", 2, 4096, 20, False, 32),
            "c3": ("headline_c3_rigid", 8, 2048, 30, True, 64),
            # round 6: configs[0] in full -- the reference's own CPU-runnable case (demo pair: 2 views x 1024 points, rap_12, all 10
            # steps, rigidity forcing on), input seed 2024; every 8th point of every step is kept
            "c0": ("headline_c0_rigid", 2, 1024, 10, True, 8, 2024),
            # round 3: configs[4] geometry through ALL 12 layers and two re-noised flow steps with rigidity
            # forcing (attention at L = 65 536), and the first pair of RANK 1 of the configs[2] job (bench.py: rank r owns
            # samples [32 r, 32 r + 32), sample b is seeded 1234 + b)
            "c4_steps": ("headline_c4_steps", 2, 32768, 2, True, 64, 1234),
            "c2_rank1": ("headline_c2_rank1", 2, 4096, 20, True, 32, 1234 + 32),
            # round 3, full-size cases beyond the uniform BASELINE geometries: a RAGGED 4-part sample (three views of unequal size and a
            # trailing empty part) through rap_12, and the configs[1] pair through the reference's largest model (rap_16)
            "c1_ragged": ("headline_c1_ragged", None, None, 20, True, 32, 4321, [[4096, 2500, 1000, 0]], 12),
            "c1_rap16": ("headline_c1_rap16", 2, 4096, 20, True, 32, 1234, None, 16)}
    for key in which:
        if key not in jobs:
            continue
        name, views, points, steps, rigid, stride = jobs[key][:6]
        iseed = jobs[key][6] if len(jobs[key]) > 6 else 1234
        parts = jobs[key][7] if len(jobs[key]) > 7 else None
        layers = jobs[key][8] if len(jobs[key]) > 8 else 12
        cfg_j, sd_j = cfg, sd
        if layers != 12:
            cfg_j = dict(S.RAP_12); cfg_j["num_layers"] = layers
            sd_j = S.make_weights(cfg_j, 0)
        inp = S.make_inputs(parts, seed=iseed) if parts else S.make_uniform_inputs(1, views, points, seed=iseed)
        t0 = time.perf_counter()
        ref = ref_loader.reference_sample(cfg_j, sd_j, inp, steps, rigid)
        dt = time.perf_counter() - t0
        out = {"num_layers": np.int64(layers), "weight_seed": np.int64(0), "input_seed": np.int64(iseed),
               "views": np.int64(views if views else len(parts[0])), "points": np.int64(points if points else 0),
               "num_steps": np.int64(steps), "rigidity": np.int64(int(rigid)),
               "weights_checksum": np.float64(weights_checksum(sd_j)), "reference_seconds": np.float64(dt),
               "reference_threads": np.int64(torch.get_num_threads())}
        if parts:
            out["parts"] = np.array(parts, dtype=np.int64)
        out.update(_traj_summary(ref, stride))
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        npts = int(inp["pointclouds"].shape[0])
        timings[name] = {"seconds": dt, "threads": torch.get_num_threads(), "points": npts, "flow_steps": steps, "num_layers": layers,
                         "points_per_s": npts / dt, "rigidity_forcing": rigid,
                         "what": "unmodified reference modules (oracle/ref_loader.reference_sample), fp32, CPU of the build container, "
                                 "all flow steps + final fit_transformations"}
        print(name, os.path.getsize(path) // 1024, "KiB", f"{dt:.1f} s", flush=True)
    if "c4" in which:
        cfg2 = dict(S.RAP_12); cfg2["num_layers"] = 2
        sd2 = S.make_weights(cfg2, 0)
        inp = S.make_uniform_inputs(1, 2, 32768, seed=1234)
        model = ref_loader.build_reference_dit(cfg2, sd2)
        cu_b, cu_p = O.prepare_cu_seqlens(inp)
        t0 = time.perf_counter()
        with torch.inference_mode():
            fw = model(x=inp["x_1"], timesteps=torch.tensor([0.5]), cond_coord=inp["pointclouds"], local_features=inp["features"],
                       latent_features=None, scales=inp["scales"], anchor_indices=inp["anchor_indices"],
                       cu_seqlens_batch=cu_b, cu_seqlens_part=cu_p)
        dt = time.perf_counter() - t0
        v = fw["velocity"] if isinstance(fw, dict) else fw
        path = os.path.join(GOLDEN_DIR, "headline_c4_forward.npz")
        np.savez_compressed(path, num_layers=np.int64(2), weight_seed=np.int64(0), input_seed=np.int64(1234), views=np.int64(2),
                            points=np.int64(32768), timestep=np.float32(0.5), weights_checksum=np.float64(weights_checksum(sd2)),
                            velocity=v.numpy(), reference_seconds=np.float64(dt))
        timings["headline_c4_forward"] = {"seconds": dt, "threads": torch.get_num_threads(), "points": 65536, "layers": 2}
        print("headline_c4_forward", os.path.getsize(path) // 1024, "KiB", f"{dt:.1f} s", flush=True)
    prof = os.path.join(os.path.dirname(GOLDEN_DIR), "..", "profiles", os.environ.get("RAP_GOLDEN_TIMINGS", "r03_cpu_reference_headline.json"))
    prof = os.path.normpath(prof)
    old = {}
    if os.path.exists(prof):
        with open(prof) as f:
            old = json.load(f)
    old.update(timings)
    with open(prof, "w") as f:
        json.dump(old, f, indent=1)


if __name__ == "__main__":
    only = [a for a in sys.argv[1:] if a.endswith("-only")]     # e.g. --overlap-only regenerates one fixture
    if "--headline-only" in only:                               # ~1.5 h of CPU: never part of the default regeneration
        make_headline_goldens(tuple(a[2:] for a in sys.argv[1:] if a[2:] in ("c0", "c1_rigid", "c1_free", "c3", "c4", "c4_steps", "c2_rank1", "c1_ragged", "c1_rap16")) or ("c1_rigid", "c1_free", "c3", "c4"))
        sys.exit(0)
    if any(a.startswith("--case=") for a in sys.argv[1:]):      # only the named sampler fixtures (main() filters)
        main()
        sys.exit(0)
    if not only:
        main()
    if not only or "--selection-only" in only:
        make_selection_golden()
    if not only or "--transforms-only" in only:
        make_transform_golden()
    if not only or "--overlap-only" in only:
        make_overlap_golden()
    if not only or "--envelope-only" in only:
        make_envelope_goldens()
    if not only or "--transform-errors-only" in only:
        make_transform_errors_golden()
    if not only or "--spinnet-only" in only:
        make_spinnet_golden()
    if not only or "--nn-only" in only:
        make_nn_metrics_golden()
    if not only or "--voxel-only" in only:
        make_voxel_golden()
    if not only or "--collate-only" in only:
        make_collate_golden()
