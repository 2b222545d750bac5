#!/usr/bin/env python3
"""n_generations as ONE rap_sample call vs the reference's sequential loop (reference modeling.py:351-361).
One pair of 2 x 1024 points (configs[0] geometry, the shipped batch_size: 1), 10 flow steps, rap_12, G = 1 / 4 generations incl. the
rigidity-based selection, fp32 and bf16.  Prints one JSON line per measurement."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rap_amd
from rap_amd import synthetic as S

dev = torch.device("cuda:0")
cfg = dict(S.RAP_12)
sd = S.make_weights(cfg, 0)
for views, points, steps in ((2, 1024, 10), (2, 4096, 20)):
    inp = S.make_uniform_inputs(1, views, points, seed=1234)
    d = {k: v.to(dev) for k, v in inp.items()}
    for dtype in ("float32", "bfloat16"):
        m = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=12, num_heads=8, local_feat_dim=32, attn_dtype=dtype,
                                  compute_dtype=dtype)
        m.load_state_dict(sd); m.to(dev)
        res = {}
        for G, batched in ((1, True), (4, True), (4, False)):
            flow = rap_amd.RectifiedPointFlow(flow_model=m, inference_sampling_steps=steps, rigidity_forcing=True, n_generations=G)
            x1s = [torch.randn(d["x_1"].shape, generator=torch.Generator().manual_seed(100 + g)).to(dev) for g in range(G)]
            for _ in range(3):
                flow.sample_generations(d, x_1_list=x1s, batch_generations=batched)
            torch.cuda.synchronize()
            n = 10
            t0 = time.perf_counter()
            for _ in range(n):
                out = flow.sample_generations(d, x_1_list=x1s, batch_generations=batched)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / n
            res[(G, batched)] = ms
            print(json.dumps({"geometry": f"1 sample x {views} x {points}, {steps} steps, rap_12", "dtype": dtype, "n_generations": G,
                              "one_call": batched and G > 1, "ms_per_sample_generations_call": ms,
                              "best": out["best_gen_indices"].tolist()}), flush=True)
        print(json.dumps({"geometry": f"1 sample x {views} x {points}", "dtype": dtype,
                          "four_generations_one_call_over_one_generation": res[(4, True)] / res[(1, True)],
                          "four_generations_loop_over_one_generation": res[(4, False)] / res[(1, True)]}), flush=True)
