#!/usr/bin/env python3
"""Few-token latency A/B (round 6): one sampling call at a time, un-instrumented, for the call sizes the reference's users live at
(config/RAP_inference.yaml:30-36: batch_size 1).  Geometries: configs[0] (1 pair x 2 x 1024, 10 steps) and one sample of 8 views x N points,
20 steps.  Variants = tuning keys 18 (four-stage GEMM ring up to this many blocks), 19 (combine + LayerNorm fusion), 7 (fused qk-norm), 20 (16-bit
attention: small work items + four-stage ring).
JSON lines on stdout.  usage: small_call_latency.py [--modes=bfloat16,float32x2,float32] [--variants=r6,r5,...] [--sizes=1000,2000]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rap_amd
from rap_amd import _lib, synthetic as S


def arg(name, default):
    for a in sys.argv[1:]:
        if a.startswith(f"--{name}="):
            return a.split("=", 1)[1].split(",")
    return default


VARIANTS = {"r6": {18: 256, 19: 1, 7: 1, 20: 1}, "r5": {18: 0, 19: 0, 7: 0, 20: 0}, "ring-only": {18: 256, 19: 0, 7: 1, 20: 0}, "fused-only": {18: 0, 19: 1, 7: 1, 20: 0},
            "no-attn-small": {18: 256, 19: 1, 7: 1, 20: 0}, "bq64": {18: 256, 19: 1, 7: 1, 20: 64}, "bq128": {18: 256, 19: 1, 7: 1, 20: 128}, "ring512": {18: 512, 19: 1, 7: 1, 20: 1}, "ring1024": {18: 1024, 19: 1, 7: 1, 20: 1}, "ring4096": {18: 4096, 19: 1, 7: 1, 20: 1},
            "unfused-qknorm": {18: 256, 19: 1, 7: 0, 20: 1},
            # 16-bit attention: no key groups (the first form of round 6) / forced 64 rows x 4 key groups / forced 128 rows x 2 key groups
            "no-kgroups": {18: 256, 19: 1, 7: 1, 20: 2}, "kg4": {18: 256, 19: 1, 7: 1, 20: 66}, "kg2": {18: 256, 19: 1, 7: 1, 20: 130}}
dev = torch.device("cuda:0")
lib = _lib.load()
cfg = dict(S.RAP_12)
sd = S.make_weights(cfg, 0)
assert lib.rap_set_tuning(17, 0) == 0
geoms = [("configs[0] geometry: 1 pair x 2 x 1024, 10 steps", [[1024, 1024]], 10)]
for n in [int(x) for x in arg("sizes", [])]:
    geoms.append((f"1 sample x 8 x {n}, 20 steps", [[n] * 8], 20))
for dtype in arg("modes", ["bfloat16", "float32x2", "float32"]):
    m = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=12, num_heads=8, local_feat_dim=32, attn_dtype=dtype, compute_dtype=dtype)
    m.load_state_dict(sd); m.to(dev)
    for label, parts, steps in geoms:
        flow = rap_amd.RectifiedPointFlow(flow_model=m, inference_sampling_steps=steps, rigidity_forcing=True)
        d = {k: v.to(dev) for k, v in S.make_inputs(parts, seed=1234).items()}
        for var in arg("variants", ["r6", "r5"]):
            for k, v in {**VARIANTS["r6"], **VARIANTS[var]}.items():      # the defaults first: a variant names only what it changes
                assert lib.rap_set_tuning(k, v) == 0
            for _ in range(5):
                flow.sample_and_register(d, x_1=d["x_1"])
            torch.cuda.synchronize()
            reps = 20 if sum(map(sum, parts)) <= 16384 else 5
            t0 = time.perf_counter()
            for _ in range(reps):
                flow.sample_and_register(d, x_1=d["x_1"])
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / reps
            print(json.dumps({"dtype": dtype, "geometry": label, "tokens": sum(map(sum, parts)), "variant": var, "tuning": VARIANTS[var], "ms_per_call": ms}), flush=True)
for k, v in VARIANTS["r6"].items():
    lib.rap_set_tuning(k, v)
lib.rap_set_tuning(17, 1024)
