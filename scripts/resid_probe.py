#!/usr/bin/env python3
"""What the fp32 residual read-modify-write costs the two N = 512 GEMMs of a 16-bit layer (round 3 probe, before building a 16-bit
residual stream): the same GEMM with (a) fp32 residual in + fp32 out (shipped), (b) fp32 out only, (c) 16-bit out only.  JSON lines."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rap_amd import _lib  # noqa: E402
from kernel_bench import timeit  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
st = lambda: _lib.current_stream(dev)  # noqa: E731
g = torch.Generator(device=dev).manual_seed(0)
TP, N = 262144, 512
for K in (512, 2048):
    A = torch.randn(TP, K, device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    Cf = torch.zeros(TP, N, device=dev)
    Ch = torch.zeros(TP, N, device=dev, dtype=torch.bfloat16)
    Cq = torch.zeros(TP, N, device=dev, dtype=torch.float16)
    cases = {"fp32 resid in + fp32 out (shipped)": (1, Cf, Cf), "fp32 out, no residual": (1, Cf, None), "16-bit out, no residual": (0, Ch, None),
             "fp16 resid in + fp16 out (epilogue 6)": (6, Cq, Cq)}
    for name, (epi, C, resid) in cases.items():
        def fn():
            rc = lib.rap_gemm_h16(1, epi, _lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(C), N, TP, N, K, _lib.ptr(bias), _lib.ptr(resid),
                                  N if resid is not None else 0, 0, _lib.ptr(None), 0, st())
            assert rc == 0, rc
        t = timeit(fn, iters=10, warm=3)
        print(json.dumps({"K": K, "case": name, "ms": round(t * 1e3, 4), "tflops": round(2.0 * TP * N * K / t / 1e12, 1)}), flush=True)
