#!/usr/bin/env python3
"""Whole-call efficiency against the SIZE of a call (round 4): the reference ships batch_size: 1 with 2 ... 512 parts of 200 ... 40 000
points (config/RAP_inference.yaml:30-36), so single-sample calls of 10^4 ... 4 x 10^5 tokens are its everyday regime, not the 262 144
tokens of BASELINE configs[1].  One sample of 8 views x N points, 20 flow steps, rap_12, rigidity on; algorithmic TFLOP/s of the whole
call (bench.call_flops) in fp32 and bf16.  JSON lines."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import rap_amd
from rap_amd import synthetic as S

dev = torch.device("cuda:0")
cfg = dict(S.RAP_12)
sd = S.make_weights(cfg, 0)
sizes = [int(x) for x in sys.argv[1:] if not x.startswith("--")] or [1000, 2000, 4000, 8000, 16000, 32000]
modes = [a.split("=", 1)[1].split(",") for a in sys.argv[1:] if a.startswith("--modes=")]
if "--x2-forced" in sys.argv:          # split precision at EVERY size (default: calls below 4 096 token rows run the exact-fp32 kernels)
    from rap_amd import _lib
    assert _lib.load().rap_set_tuning(17, 0) == 0
for dtype in (modes[0] if modes else ("float32", "bfloat16")):
    m = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=12, num_heads=8, local_feat_dim=32, attn_dtype=dtype, compute_dtype=dtype)
    m.load_state_dict(sd); m.to(dev)
    flow = rap_amd.RectifiedPointFlow(flow_model=m, inference_sampling_steps=20, rigidity_forcing=True)
    for n in sizes:
        parts = [[n] * 8]
        inp = S.make_inputs(parts, seed=7)
        d = {k: v.to(dev) for k, v in inp.items()}
        flops = bench.call_flops(parts, 12, 20)
        flow.sample_and_register(d, x_1=d["x_1"]); torch.cuda.synchronize()
        reps = 3 if flops < 5e14 else 1
        t0 = time.perf_counter()
        for _ in range(reps):
            flow.sample_and_register(d, x_1=d["x_1"])
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / reps
        print(json.dumps({"dtype": dtype, "tokens": 8 * n, "geometry": f"1 sample x 8 x {n}", "ms_per_call": 1e3 * t, "points_per_s": 8 * n / t,
                          "algorithmic_tflop": flops / 1e12, "achieved_tflops_whole_call": flops / t / 1e12,
                          "attention_share_of_flops": 1 - (8 * n * (12 * bench.DENSE_FLOPS_PER_TOKEN_LAYER + bench.EMBED_HEAD_FLOPS_PER_TOKEN) * 20) / flops}), flush=True)
