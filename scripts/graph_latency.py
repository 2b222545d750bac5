#!/usr/bin/env python3
"""Latency of one sampling call at demo size (BASELINE configs[0] geometry: 1 pair, 2 x 1024 points, 10 steps, rap_12), enqueued
launch by launch vs replayed as one captured HIP graph.  Prints one JSON line per (dtype, mode)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rap_amd
from rap_amd import synthetic as S

dev = torch.device("cuda:0")
cfg = dict(S.RAP_12)
sd = S.make_weights(cfg, 0)
for dtype in ("float32", "bfloat16"):
    m = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=cfg["num_layers"], num_heads=8, local_feat_dim=32,
                              attn_dtype="float32", compute_dtype=dtype)
    m.load_state_dict(sd); m.to(dev)
    for pairs, pts, steps in ((1, 1024, 10), (1, 4096, 20), (4, 4096, 20)):
        flow = rap_amd.RectifiedPointFlow(flow_model=m, inference_sampling_steps=steps, rigidity_forcing=True)
        inp = {k: v.to(dev) for k, v in S.make_inputs([[pts, pts]] * pairs, seed=1).items()}
        flow.sample_and_register(inp, x_1=inp["x_1"]); torch.cuda.synchronize()
        def timed(fn, n=5):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        eager = timed(lambda: flow.sample_and_register(inp, x_1=inp["x_1"]))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = flow.sample_and_register(inp, x_1=inp["x_1"])
        graph = timed(g.replay)
        print(json.dumps({"dtype": dtype, "pairs": pairs, "points_per_view": pts, "steps": steps, "eager_ms": round(eager, 3),
                          "graph_ms": round(graph, 3), "speedup": round(eager / graph, 3)}), flush=True)
