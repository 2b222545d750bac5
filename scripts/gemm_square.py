#!/usr/bin/env python3
"""16-bit GEMM (epilogue 0: bf16 out = acc + bias) on the guide's benchmark shapes (4096^3, 8192^3) next to this model's shapes: is the
k-loop of gemm_h16.hip at the level of the guide's 256^2 8-phase template (1 320 TF @4k, 1 470 @8k on random operands), i.e. is what the
model's GEMMs lose a property of their SHAPES (K = 512: eight k-tiles per output tile, M = 262 144 rows streamed once)?"""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rap_amd import _lib
dev = torch.device("cuda:0")
lib = _lib.load()
variants = [int(v) for v in sys.argv[1:]] or [1, 0]      # rap_set_tuning(11, .): 1 = persistent kernel where the shape allows (>= 512 full tiles), 0 = one tile per block
g = torch.Generator(device=dev).manual_seed(0)
st = torch.cuda.current_stream(dev).cuda_stream
for var in variants:
    assert lib.rap_set_tuning(11, var) == 0
    for (M, N, K) in ((4096, 4096, 4096), (8192, 8192, 8192), (16384, 4096, 512), (262144, 4096, 512), (262144, 512, 2048), (32768, 512, 8192), (262144, 512, 512)):
        A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        W = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        def fn():
            rc = lib.rap_gemm_h16(1, 0, _lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(C), N, M, N, K, _lib.ptr(None), _lib.ptr(None), 0, 8, _lib.ptr(None), 0, st)
            assert rc == 0, rc
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 10
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(json.dumps({"persistent": bool(var) and M % 256 == 0 and (M // 256) * (N // 256) >= 512, "M": M, "N": N, "K": K, "ms": round(ms, 4), "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}), flush=True)
        del A, W, C
assert lib.rap_set_tuning(11, 1) == 0
