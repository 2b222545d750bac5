#!/usr/bin/env python3
"""Every epilogue of the 16-bit GEMM at the model's shapes (TP = 262144 tokens), persistent kernel on / off (rap_set_tuning key 11)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rap_amd import _lib  # noqa: E402
from kernel_bench import timeit  # noqa: E402

lib = _lib.load(); dev = torch.device("cuda:0")
st = lambda: _lib.current_stream(dev)  # noqa: E731
g = torch.Generator(device=dev).manual_seed(0)
TP, H = 262144, 8
nblk = TP // 64
dt = 1
for name, epi, N, K in (("qkv + qk-norm", 5, 1536, 512), ("out-proj fp16 stream", 7, 512, 512), ("ff1 GEGLU", 3, 4096, 512), ("ff2 fp16 stream", 7, 512, 2048),
                        ("out-proj fp32 stream", 1, 512, 512), ("ff2 fp32 stream", 1, 512, 2048)):
    A = torch.randn(TP, K, device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    gq = torch.ones(H, 64, device=dev)
    if epi == 5:
        C = torch.zeros(2, H, TP, 64, device=dev, dtype=torch.bfloat16); vt = torch.zeros(H, nblk, 64, 64, device=dev, dtype=torch.bfloat16)
        fn = lambda: lib.rap_gemm_h16_qkvnorm(dt, _lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(C), TP, K, H, _lib.ptr(gq), _lib.ptr(gq), 8.0, _lib.ptr(vt), nblk, st())  # noqa: E731
    else:
        Cw = N // 2 if epi == 3 else N
        C = torch.zeros(TP, Cw, device=dev, dtype={1: torch.float32, 7: torch.float16, 3: torch.bfloat16}[epi])
        resid = C if epi in (1, 7) else None
        fn = lambda: lib.rap_gemm_h16(dt, epi, _lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(C), Cw, TP, N, K, _lib.ptr(bias), _lib.ptr(resid), Cw if resid is not None else 0, 0, _lib.ptr(None), 0, st())  # noqa: E731
    row = {"gemm": name, "N": N, "K": K}
    for pz in (1, 0):
        assert lib.rap_set_tuning(11, pz) == 0
        assert fn() == 0
        t = timeit(lambda: fn(), iters=10, warm=3)
        row["persistent_ms" if pz else "one_tile_per_block_ms"] = round(t * 1e3, 4)
        row["persistent_tflops" if pz else "one_tile_per_block_tflops"] = round(2.0 * TP * N * K / t / 1e12, 1)
    assert lib.rap_set_tuning(11, 1) == 0
    print(json.dumps(row), flush=True)
