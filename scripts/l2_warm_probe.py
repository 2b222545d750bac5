#!/usr/bin/env python3
"""Probe (round 6): how much of a few-token GEMM launch is the fetch of operands that are NOT in the XCD's L2?

In a sampling call a layer's weights are touched once per flow step (6 MB per layer in bf16, 72 MB per step: they live in the 256 MB
Infinity Cache, not in the 8 x 4 MB of L2), and the activations a GEMM reads were written by the previous kernel from other XCDs.  The
probe times the layer's four bf16 GEMM shapes at a demo pair's row count (2 048) through the kernel-level entry rap_gemm_h16, back to back:
  warm : the same A / W / C buffers every launch (everything the launch reads sits in L2 after the first one)
  cold : a ring of N different A / W / C buffers, N x bytes > 32 MB (every launch reads operands last touched N launches ago)
One JSON line per shape: us per launch warm / cold (HIP events around ONE replay of a hipGraph of `reps` launches).
Usage (GPU box):  python scripts/l2_warm_probe.py [--rows=2048] [--reps=400]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rap_amd import _lib  # noqa: E402


def arg(name, default):
    for a in sys.argv[1:]:
        if a.startswith(f"--{name}="):
            return int(a.split("=", 1)[1])
    return default


dev = torch.device("cuda:0")
lib = _lib.load()
M, reps = arg("rows", 2048), arg("reps", 400)
# (name, N, K, epilogue): plain 16-bit output (0) for the projections, fp32 residual output (1) for the residual GEMMs
SHAPES = [("out-projection", 512, 512, 1), ("qkv (plain store)", 1536, 512, 0), ("ff1 (no GEGLU)", 4096, 512, 0), ("ff2", 512, 2048, 1)]
for name, N, K, epi in SHAPES:
    bytes_per = (M * K + N * K) * 2 + M * N * (4 if epi == 1 else 2)
    for label, nbuf in (("warm", 1), ("cold", max(2, (96 << 20) // bytes_per))):
        A = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(nbuf)]
        W = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(nbuf)]
        C = [torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == 1 else torch.bfloat16) for _ in range(nbuf)]
        bias = torch.zeros(N, device=dev)

        def launch(i):
            j = i % nbuf
            rc = lib.rap_gemm_h16(1, epi, _lib.ptr(A[j]), K, _lib.ptr(W[j]), K, _lib.ptr(C[j]), N, M, N, K, _lib.ptr(bias),
                                  _lib.ptr(C[j]) if epi == 1 else None, N if epi == 1 else 0, 0, None, 0, _lib.current_stream(dev))
            assert rc == 0, rc
        for i in range(2 * nbuf + 8):
            launch(i)
        torch.cuda.synchronize()
        # the launches are replayed from a hipGraph (as rap_sample's flow steps are): the host's launch rate is not what is measured
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for i in range(reps):
                launch(i)
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        print(json.dumps({"gemm": name, "M": M, "N": N, "K": K, "operands": label, "buffers": nbuf, "MB_per_launch": round(bytes_per / 2 ** 20, 2),
                          "us_per_launch": round(1e3 * e0.elapsed_time(e1) / reps, 2)}), flush=True)
        del A, W, C
