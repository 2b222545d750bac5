#!/usr/bin/env python3
"""Host time of ONE sampling call on an idle device: enqueued launch by launch vs replayed as a captured HIP graph (round 4: does a graph
launch avoid the HIP queue's back-pressure that makes the host follow the GPU for 7-14 % of a full-size call?).  configs[1] batch."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rap_amd
from rap_amd import synthetic as S

dev = torch.device("cuda:0")
cfg = dict(S.RAP_12)
sd = S.make_weights(cfg, 0)
for dtype in ("bfloat16", "float32"):
    m = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=12, num_heads=8, local_feat_dim=32, attn_dtype="float32", compute_dtype=dtype)
    m.load_state_dict(sd); m.to(dev)
    for pairs in (1, 32):
        flow = rap_amd.RectifiedPointFlow(flow_model=m, inference_sampling_steps=20, rigidity_forcing=True)
        inp = {k: v.to(dev) for k, v in S.make_inputs([[4096, 4096]] * pairs, seed=1).items()}
        flow.sample_and_register(inp, x_1=inp["x_1"]); torch.cuda.synchronize()
        def host_and_total(fn):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            return 1e3 * (t1 - t0), 1e3 * (t2 - t0)
        eh, et = host_and_total(lambda: flow.sample_and_register(inp, x_1=inp["x_1"]))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = flow.sample_and_register(inp, x_1=inp["x_1"])
        g.replay(); torch.cuda.synchronize()
        gh, gt = host_and_total(g.replay)
        print(json.dumps({"dtype": dtype, "pairs": pairs, "eager_host_ms": round(eh, 2), "eager_total_ms": round(et, 2), "graph_host_ms": round(gh, 2),
                          "graph_total_ms": round(gt, 2)}), flush=True)
