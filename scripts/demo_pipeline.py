#!/usr/bin/env python3
"""End-to-end walk through everything rap_amd replaces, in the order the reference's demo.py runs it, on synthetic scans
(no dataset / checkpoint offline): dense raw scans -> voxel down-sampling -> farthest point sampling -> MiniSpinNet descriptors
-> n generations of the rectified-flow sampler -> rigidity / overlap-ratio selection -> per-view transform files.  Prints one
JSON line with the stage timings.

    python scripts/demo_pipeline.py [--views 2] [--points 4096] [--generations 3] [--dtype bfloat16] [--out /tmp/rap_demo]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rap_amd  # noqa: E402
from rap_amd import synthetic as S  # noqa: E402
from rap_amd.evaluator import save_transformation_files  # noqa: E402
from rap_amd.point_sampling import sample_farthest_points, voxel_down_sample_torch  # noqa: E402
from rap_amd.spinnet import MiniSpinNet, make_spinnet_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=4)
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--generations", type=int, default=3)
    ap.add_argument("--flow-steps", type=int, default=20)
    ap.add_argument("--dtype", default="bfloat16", choices=["float32", "bfloat16", "float16"])
    ap.add_argument("--raw-factor", type=int, default=8, help="raw scan = this many jittered copies of every view point")
    ap.add_argument("--voxel", type=float, default=0.004, help="voxel size of the down-sampling (normalised units)")
    ap.add_argument("--out", default="/tmp/rap_demo")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    sync = torch.cuda.synchronize
    inp = S.make_inputs([[args.points] * args.views for _ in range(args.pairs)], seed=7)
    data = {k: v.to(dev) for k, v in inp.items()}
    B, P = inp["points_per_part"].shape

    # 0. preprocessing (extract_sample_features.py:378-470): every view arrives as a dense raw scan (here: jittered copies of the
    # synthetic view), is voxel down-sampled (dataset_utils.py:279-322) and reduced to its n points by farthest point sampling
    # (point_sampling_utils.py:263-305); the sampled points replace the view, the down-sampled cloud is MiniSpinNet's support.
    g = torch.Generator(device=dev).manual_seed(3)
    t0 = time.perf_counter()
    supports, ns, raw_points = [], [], 0
    off = 0
    for b in range(B):
        for p in range(P):
            n = int(inp["points_per_part"][b, p])
            view = data["pointclouds"][off:off + n]
            raw = view.repeat_interleave(args.raw_factor, 0) + 0.002 * torch.randn(n * args.raw_factor, 3, device=dev, generator=g)
            supports.append(raw[voxel_down_sample_torch(raw, args.voxel)])
            ns.append(n); raw_points += raw.shape[0]
            off += n
    kept_points = sum(d.shape[0] for d in supports)
    # ONE batched FPS over all views (a block per cloud): zero-padded batch + lengths + per-cloud K, as apply_batched_fps does
    lens = torch.tensor([d.shape[0] for d in supports])
    padded = torch.zeros(len(supports), int(lens.max()), 3, device=dev)
    for i, d in enumerate(supports):
        padded[i, : d.shape[0]] = d
    sampled, _ = sample_farthest_points(padded, lengths=lens, K=torch.minimum(torch.tensor(ns), lens), random_start_point=True)
    keys = []
    for i, (d, n) in enumerate(zip(supports, ns)):
        k = sampled[i, : min(n, d.shape[0])]
        keys.append(k if k.shape[0] == n else torch.cat([k, d[: n - k.shape[0]]]))     # a very coarse grid: pad with the first points
    data["pointclouds"] = torch.cat(keys)
    sync(); t_pre = time.perf_counter() - t0

    # 1. local features: MiniSpinNet descriptors of the sampled points over the down-sampled support (extract_sample_features.py:151-220)
    spin = MiniSpinNet(des_r=0.2); spin.load_state_dict(make_spinnet_weights(0)); spin.to(dev)
    t0 = time.perf_counter()
    feats = [spin(ds[None], key[None], 0.2, True, perm=np.arange(ds.shape[0]))["desc"] for ds, key in zip(supports, keys)]
    data["features"] = torch.cat(feats)
    sync(); t_feat = time.perf_counter() - t0

    # 2. n generations of the sampler + rigidity selection (test_step, modeling.py:397-592)
    cfg = dict(S.RAP_12)
    model = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=12, num_heads=8, local_feat_dim=32,
                                  compute_dtype=args.dtype)
    model.load_state_dict(S.make_weights(cfg, 0)); model.to(dev)
    flow = rap_amd.RectifiedPointFlow(flow_model=model, inference_sampling_steps=args.flow_steps, rigidity_forcing=True,
                                      n_generations=args.generations)
    sync(); t0 = time.perf_counter()
    out = flow.sample_generations(data)
    sync(); t_sample = time.perf_counter() - t0

    # 3. second criterion: overlap ratio of every generation's final cloud (modeling.py:594-601)
    t0 = time.perf_counter()
    ov = torch.stack([rap_amd.compute_overlap_ratio(g["end_point_trajectory"][-1], inp["points_per_part"], inp["cu_seqlens"], [0.01])[0]
                      for g in out["generations"]])
    best_ov, _, _, _ = rap_amd.select_generations_by_overlap(ov, torch.stack([g["end_point_trajectory"][-1] for g in out["generations"]]),
                                                             torch.stack([g["R"] for g in out["generations"]]),
                                                             torch.stack([g["t"] for g in out["generations"]]), inp["cu_seqlens"])
    sync(); t_overlap = time.perf_counter() - t0

    # 4. transform files of the rigidity-selected generation (evaluator.py:383-490); synthetic GT = identity
    gt = {"rotations": torch.eye(3).expand(B, P, 3, 3).contiguous(), "translations": torch.zeros(B, P, 3), "scales": inp["scales"],
          "points_per_part": inp["points_per_part"]}
    t0 = time.perf_counter()
    paths = save_transformation_files(gt, args.out, "synthetic", list(range(B)), "selected", out["rotations_selected"],
                                      out["translations_selected"])
    t_files = time.perf_counter() - t0
    print(json.dumps({"pairs": B, "views": P, "points_per_view": args.points, "generations": args.generations, "dtype": args.dtype,
                      "raw_points": raw_points, "points_after_voxel_downsampling": kept_points,
                      "seconds": {"voxel_downsample_and_fps": t_pre, "miniSpinNet_features": t_feat, "sample_generations_incl_rigidity_selection": t_sample,
                                  "overlap_ratio_selection": t_overlap, "transform_files": t_files},
                      "best_by_rigidity": out["best_gen_indices"].tolist(), "best_by_overlap": best_ov.tolist(),
                      "rigidity_rmse_m": out["rigidity_rmse"].cpu().tolist(), "files_written": len(paths), "out_dir": args.out}))


if __name__ == "__main__":
    main()
