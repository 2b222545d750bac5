"""End-to-end on the device, through rap_amd only: raw multi-view scans in a WORLD frame (metres) -> statistical outlier removal ->
voxel down-sampling -> voxel-adaptive sample counts -> batched FPS -> MiniSpinNet descriptors -> transform_and_collate (centring,
scale normalisation, anchor) -> rectified-flow sampler with rigidity forcing -> per-part `*_transform.txt` files.

No golden is needed for the whole chain (each stage has its own parity test against the reference); what is asserted here is the
size-independent contract BETWEEN the stages, the one `demo.py:1332-1342` relies on when it applies the written matrices to the
original scans: a file transform maps the RAW part into the metric, globally centred frame in which the predicted cloud lives,

    T_file[p] . X_raw[p]  ==  scale * (cond[p] R_pred[p]^T + t_pred[p])          for every part p of every sample,

(checked on the CPU with the unmodified reference collate + the oracle's writer: 2e-6 m at 100 m offsets), so that
`inv(T_file[0]) T_file[p]` registers the original scans exactly as the sampler registered the normalised ones.
"""
import numpy as np
import pytest
import torch

import rap_amd
from rap_amd import synthetic as S
from rap_amd.data import transform_and_collate
from rap_amd.evaluator import save_transformation_files
from rap_amd.point_sampling import (calculate_adaptive_sample_count_per_part, remove_statistical_outlier, sample_farthest_points,
                                    voxel_down_sample_torch)
from rap_amd.spinnet import MiniSpinNet, make_spinnet_weights

pytestmark = pytest.mark.gpu


def _rot(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q); w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _raw_scans(rng, n_views, pts_per_view, origin):
    """views of one scene: overlapping slabs of a box surface cloud, each in the SAME world frame, metres, far from the origin,
    plus a few far outliers per view"""
    scene = rng.uniform(-4, 4, size=(n_views * pts_per_view, 3)) * np.array([1.0, 1.0, 0.25])
    scene = scene @ _rot(rng).T + origin
    views = []
    for v in range(n_views):
        lo = int(v * 0.6 * pts_per_view)
        view = scene[lo:lo + pts_per_view].copy()
        out = view[:6] + rng.normal(size=(6, 3)) * 30.0           # isolated far points: statistical outliers
        views.append(np.concatenate([view, out]).astype(np.float32))
    return views


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_raw_scans_to_transform_files(dev, tmp_path):
    rng = np.random.default_rng(5)
    B, P = 2, 3
    raw = [_raw_scans(rng, P, 1500, np.array([120.0, -60.0, 8.0]) * (b + 1)) for b in range(B)]

    # ---- preprocessing of every view (extract_sample_features.py:378-470)
    spin = MiniSpinNet(des_r=0.6); spin.load_state_dict(make_spinnet_weights(0)); spin.to(dev)
    samples, kept_raw = [], []
    for b in range(B):
        clean = []
        for v in raw[b]:
            pts, idx = remove_statistical_outlier(torch.from_numpy(v).to(dev), 20, 2.5)
            assert pts.shape[0] <= v.shape[0] - 4, "the planted far points must go"          # at least most of the 6 outliers
            clean.append(pts)
        counts = calculate_adaptive_sample_count_per_part(clean, voxel_size=0.25, voxel_ratio=0.5, min_points_per_part=64,
                                                          max_sample_points=2000)
        assert len(counts) == P and all(64 <= c <= pts.shape[0] for c, pts in zip(counts, clean))
        parts, feats = [], []
        for pts, n in zip(clean, counts):
            ds = pts[voxel_down_sample_torch(pts, 0.05)]
            k = min(int(n), ds.shape[0])
            sampled, _ = sample_farthest_points(ds[None], K=k, random_start_point=False)
            key = sampled[0, :k]
            desc = spin(ds[None], key[None], 0.6, True, perm=np.arange(ds.shape[0]))["desc"].reshape(k, 32)
            assert torch.isfinite(desc).all()
            assert (desc.norm(dim=1) - 1).abs().max().item() < 1e-4                          # patch_embedder.py:83: unit descriptors
            parts.append(key); feats.append(desc)
        samples.append({"parts": parts, "features": feats})
        kept_raw.append([p.clone() for p in parts])

    # ---- the boundary batch, the sampler, the poses
    batch = transform_and_collate(samples, max_parts=4, shuffle=False)
    ppp = batch["points_per_part"]
    assert ppp.shape == (B, 4) and int(ppp.sum()) == batch["pointclouds"].shape[0]
    cfg = dict(S.RAP_12); cfg["num_layers"] = 2
    model = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=2, num_heads=8, local_feat_dim=32, attn_dtype="float32")
    model.load_state_dict(S.make_weights(cfg, 3)); model.to(dev)
    flow = rap_amd.RectifiedPointFlow(flow_model=model, inference_sampling_steps=4, rigidity_forcing=True, validate_inputs=True)
    out = flow.sample_and_register(batch, x_1=torch.randn(batch["pointclouds"].shape, generator=torch.Generator().manual_seed(1)).to(dev))
    R, t = out["R"], out["t"]
    assert torch.isfinite(R).all() and torch.isfinite(t).all()

    # ---- the files, as the reference writes them (evaluator.py:383-490) ...
    paths = save_transformation_files(batch, tmp_path, "synthetic", list(range(B)), 0, R, t,
                                      global_rotation=batch["global_rotation"], global_translation=batch["global_translation"])
    assert len(paths) == int((ppp > 0).sum())

    # ---- ... applied to the RAW (world-frame) parts land where the sampler put the normalised ones, in metres
    cond = batch["pointclouds"].double().cpu()
    off, worst = 0, 0.0
    for b in range(B):
        s = float(batch["scales"][b])
        for p in range(4):
            n = int(ppp[b, p])
            if n == 0:
                continue
            T = np.loadtxt(tmp_path / f"synthetic_sample{b:05d}_generation00_part{p:02d}_transform.txt")
            assert T.shape == (4, 4) and np.allclose(T[3], [0, 0, 0, 1])
            X = kept_raw[b][p].double().cpu().numpy()
            Y = X @ T[:3, :3].T + T[:3, 3]
            Z = s * (cond[off:off + n].numpy() @ R[b, p].double().cpu().numpy().T + t[b, p].double().cpu().numpy())
            worst = max(worst, float(np.abs(Y - Z).max()))
            off += n
    # fp32 coordinates at |x| ~ 250 m carry 1.5e-5 m; the files 1e-8
    assert worst < 2e-3, worst
    print(f"raw scans -> files -> registered cloud: max |T_file X_raw - s (cond R^T + t)| = {worst:.2e} m")
