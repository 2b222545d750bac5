#!/usr/bin/env python3
"""Sample-call time vs the number of concurrent batch shards (RectifiedPointFlow(num_streams=n)), headline workload."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rap_amd
from rap_amd import synthetic as S

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bfloat16")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--streams", default="1,2,3,4,1")
ap.add_argument("--sequential", action="store_true", help="run the shards one after the other on one stream (cache footprint only)")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = dict(S.RAP_12)
m = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=12, num_heads=8, local_feat_dim=32, compute_dtype=args.dtype)
m.load_state_dict(S.make_weights(cfg, 0)); m.to(dev)
inp = S.make_inputs([[4096] * 2 for _ in range(args.batch)], seed=1234)
data = {k: v.to(dev) for k, v in inp.items()}
data["cu_seqlens"] = inp["cu_seqlens"]            # on the host, as the reference's collate delivers it: no read-back for the cut
pts = args.batch * 2 * 4096
ref = None
for n in [int(x) for x in args.streams.split(",")]:
    flow = rap_amd.RectifiedPointFlow(flow_model=m, inference_sampling_steps=20, rigidity_forcing=True, num_streams=n)
    flow._sequential_shards = args.sequential
    out = flow.sample_and_register(data, x_1=data["x_1"]); torch.cuda.synchronize()
    if ref is None:
        ref = out["end_point_trajectory"][-1].clone()
    dev_max = float((out["end_point_trajectory"][-1] - ref).abs().max())
    t0 = time.perf_counter()
    for _ in range(args.reps):
        out = flow.sample_and_register(data, x_1=data["x_1"])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    print(json.dumps({"dtype": args.dtype, "streams": n, "sequential": args.sequential, "ms_per_call": round(dt * 1e3, 1), "points_per_s": round(pts / dt),
                      "final_cloud_max_abs_vs_one_stream": dev_max}), flush=True)
