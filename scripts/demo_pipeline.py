#!/usr/bin/env python3
"""End-to-end walk through everything rap_amd replaces, in the order the reference's demo.py runs it, on synthetic scans
(no dataset / checkpoint offline): raw views -> MiniSpinNet descriptors -> n generations of the rectified-flow sampler ->
rigidity / overlap-ratio selection -> per-view transform files.  Prints one JSON line with the stage timings.

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
from rap_amd.spinnet import MiniSpinNet, make_spinnet_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=4)
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--generations", type=int, default=3)
    ap.add_argument("--flow-steps", type=int, default=20)
    ap.add_argument("--dtype", default="bfloat16", choices=["float32", "bfloat16", "float16"])
    ap.add_argument("--out", default="/tmp/rap_demo")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    sync = torch.cuda.synchronize
    inp = S.make_inputs([[args.points] * args.views for _ in range(args.pairs)], seed=7)
    data = {k: v.to(dev) for k, v in inp.items()}
    B, P = inp["points_per_part"].shape

    # 1. local features: MiniSpinNet on every view (the reference: extract_sample_features.py:151-220), keypoints = the points
    spin = MiniSpinNet(des_r=0.2); spin.load_state_dict(make_spinnet_weights(0)); spin.to(dev)
    t0 = time.perf_counter()
    feats = []
    off = 0
    for b in range(B):
        for p in range(P):
            n = int(inp["points_per_part"][b, p])
            view = data["pointclouds"][off:off + n]
            feats.append(spin(view[None], view[None], 0.2, True, perm=np.arange(n))["desc"])
            off += n
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
                      "seconds": {"miniSpinNet_features": t_feat, "sample_generations_incl_rigidity_selection": t_sample,
                                  "overlap_ratio_selection": t_overlap, "transform_files": t_files},
                      "best_by_rigidity": out["best_gen_indices"].tolist(), "best_by_overlap": best_ov.tolist(),
                      "rigidity_rmse_m": out["rigidity_rmse"].cpu().tolist(), "files_written": len(paths), "out_dir": args.out}))


if __name__ == "__main__":
    main()
