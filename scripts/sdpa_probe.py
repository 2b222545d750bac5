#!/usr/bin/env python3
"""Which attention evaluation should the DEVICE-side checker (oracle/rap_oracle.py, tests only) use?  Times, in fp32 on the GPU,
the explicit chunked matmul + softmax it uses now against torch's fused SDPA backends (if this torch build has an fp32 one), and
reports their max-abs difference from an fp64 evaluation on a slice of the rows."""
import json
import sys, os, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rap_oracle as O

dev = torch.device("cuda:0")
for L in (8192, 65536):
    g = torch.Generator(device=dev).manual_seed(1)
    q, k, v = (torch.randn(8, L, 64, device=dev, generator=g) for _ in range(3))
    ref = O._softmax_attention_chunked(q[:, :256].double(), k.double(), v.double()) if L <= 8192 else None
    def timeit(fn, n=3):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            o = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n, o
    t, o = timeit(lambda: O._softmax_attention_chunked(q, k, v))
    row = {"L": L, "chunked_explicit_ms": 1e3 * t}
    if ref is not None:
        # the double reference was computed with q rows 0..255 only -> compare those rows
        row["chunked_explicit_err_vs_fp64"] = float((o[:, :256].double() - ref).abs().max())
    try:
        from torch.nn.attention import sdpa_kernel, SDPBackend
        for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)):
            if name == "math" and L > 8192:
                continue
            try:
                with sdpa_kernel([be]):
                    t, o = timeit(lambda: F.scaled_dot_product_attention(q[None], k[None], v[None])[0])
                row[f"sdpa_{name}_ms"] = 1e3 * t
                if ref is not None:
                    row[f"sdpa_{name}_err_vs_fp64"] = float((o[:, :256].double() - ref).abs().max())
            except Exception as e:      # backend has no fp32 kernel in this build
                row[f"sdpa_{name}"] = f"unavailable: {type(e).__name__}: {str(e)[:120]}"
    except ImportError as e:
        row["sdpa"] = f"unavailable: {e}"
    print(json.dumps(row), flush=True)
