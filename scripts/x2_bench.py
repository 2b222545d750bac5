#!/usr/bin/env python3
"""Micro-benchmark of the split-precision ("float32x2") kernels at the BASELINE configs[1] shapes (TP = 262 144 tokens), through the C ABI,
next to the exact-fp32 kernels they replace, interleaved in one process.  Also probes whether the fp16 matrix pipe keeps SUBNORMAL
operands (the tails of small values are fp16 subnormals; the weight planes are scaled so that theirs are not).

Prints JSON lines: per kernel ms, fp32-equivalent TFLOP/s (algorithmic FLOPs / time), fraction of the split-precision peak
(2 500 / 3 TFLOP/s) and the speed-up over the fp32 kernel of the same shape.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rap_amd import _lib  # noqa: E402
from rap_amd.flow_model import workspace  # noqa: E402

PEAK_X2 = 2500.0 / 3.0
PEAK_F32 = 157.3


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=262144)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = lambda: _lib.current_stream(dev)
    g = torch.Generator(device=dev).manual_seed(0)
    TP, d, H = args.tokens, 512, 8
    nblk = (TP + 255) // 256 * 256 // 64
    rows = []

    def emit(r):
        rows.append(r)
        print(json.dumps(r), flush=True)

    def pack(x, scale=1.0):
        out = torch.empty(x.shape[0], 2 * x.shape[1], dtype=torch.float16, device=dev)
        assert lib.rap_x2_pack(_lib.ptr(x), x.shape[1], x.shape[0], x.shape[1], float(scale), _lib.ptr(out), st()) == 0
        return out

    # ---- subnormal probe: A = 1, W = 2^-20 (head is an fp16 SUBNORMAL, tail 0): the product survives only if the pipe keeps subnormals
    if args.only in ("", "probe"):
        M, N, K = 256, 256, 64
        A = torch.ones(M, K, device=dev); W = torch.full((N, K), 2.0 ** -20, device=dev)
        C = torch.zeros(M, N, device=dev)
        Ap, Wp = pack(A), pack(W)
        rc = lib.rap_x2_gemm(1, _lib.ptr(Ap), 2 * K, _lib.ptr(Wp), 2 * K, _lib.ptr(C), N, M, N, 2 * K, None, None, 0, 1.0, 0, None, None, 8.0, None, 0, st())
        assert rc == 0, rc
        torch.cuda.synchronize()
        got = float(C[0, 0]); want = K * 2.0 ** -20
        emit({"probe": "fp16 MFMA with a SUBNORMAL operand (2^-20)", "got": got, "want": want, "subnormals_kept": abs(got - want) < 1e-3 * want})

    # ---- GEMMs of one layer
    def gemm_pair(name, epi_x2, epi_f32, N, K, ldc_x2, out_cols_f32):
        A = torch.randn(TP, K, device=dev, generator=g)
        W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
        bias = torch.randn(N, device=dev, generator=g)
        gq = torch.rand(H, 64, device=dev, generator=g) + 0.5; gk = torch.rand(H, 64, device=dev, generator=g) + 0.5
        sc = 2.0 ** 14
        Ap, Wp = pack(A), pack(W, sc)
        if epi_x2 == 1:
            Cx = torch.zeros(TP, N, device=dev); args_x = (Cx, N, Cx, N)
        elif epi_x2 == 3:
            Cx = torch.zeros(TP, ldc_x2, device=dev, dtype=torch.float16); args_x = (Cx, ldc_x2, None, 0)
        else:
            Cx = torch.zeros(2 * H * 2 * TP * 64, device=dev, dtype=torch.float16); args_x = (Cx, 0, None, 0)
        vt = torch.zeros(H * nblk * 2 * 64 * 64, device=dev, dtype=torch.float16) if epi_x2 == 5 else None

        def fx():
            rc = lib.rap_x2_gemm(epi_x2, _lib.ptr(Ap), 2 * K, _lib.ptr(Wp), 2 * K, _lib.ptr(args_x[0]), args_x[1], TP, N, 2 * K, _lib.ptr(bias) if epi_x2 != 5 else None,
                                 _lib.ptr(args_x[2]), args_x[3], 1.0 / sc, H, _lib.ptr(gq), _lib.ptr(gk), 8.0, _lib.ptr(vt), nblk if epi_x2 == 5 else 0, st())
            assert rc == 0, rc
        tx = timeit(fx)
        Cf = torch.zeros(TP, out_cols_f32, device=dev)

        def ff():
            rc = lib.rap_gemm_f32(epi_f32, _lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(Cf), out_cols_f32, TP, N, K, _lib.ptr(bias) if epi_f32 != 4 else None,
                                  _lib.ptr(Cf) if epi_f32 == 1 else None, out_cols_f32 if epi_f32 == 1 else 0, None, None, H if epi_f32 == 4 else 0, st())
            assert rc == 0, rc
        tf = timeit(ff, iters=3, warm=1)
        fl = 2.0 * TP * N * K
        emit({"kernel": f"gemm[{name}]", "M": TP, "N": N, "K": K, "x2_ms": tx * 1e3, "f32_ms": tf * 1e3, "x2_tflops_fp32_equiv": fl / tx / 1e12,
              "x2_frac_of_833TF": fl / tx / 1e12 / PEAK_X2, "f32_tflops": fl / tf / 1e12, "f32_frac_of_157TF": fl / tf / 1e12 / PEAK_F32, "speedup": tf / tx})
        del A, W, Ap, Wp, Cx, Cf

    if args.only in ("", "gemm"):
        gemm_pair("qkv+qknorm", 5, 4, 3 * d, d, 0, 3 * d)       # (the fp32 side here is the plain head-major projection, without the norm)
        gemm_pair("out-proj+resid", 1, 1, d, d, 0, d)
        gemm_pair("ff1+geglu", 3, 3, 8 * d, d, 8 * d, 4 * d)
        gemm_pair("ff2+resid", 1, 1, d, 4 * d, 0, d)

    # ---- attention, per part (L = 4096) and per sample (L = 8192)
    if args.only in ("", "attn"):
        import torch.nn.functional as F
        qk = (torch.randn(2 * H * 2 * TP * 64, device=dev, generator=g) * 0.5).to(torch.float16)
        vt = torch.randn(H * nblk * 2 * 64 * 64, device=dev, generator=g).to(torch.float16)
        out = torch.zeros(TP, 2 * d, device=dev, dtype=torch.float16)
        qkv32 = torch.randn(3 * H * TP * 64, device=dev, generator=g) * 0.5
        out32 = torch.zeros(TP, d, device=dev)
        bound = torch.full((H,), 30.0, device=dev)
        for L in (4096, 8192):
            nseg = TP // L
            cu = (torch.arange(nseg + 1, dtype=torch.int32) * L).to(dev)
            ws = workspace(dev, lib.rap_attention_workspace_bytes(TP, nseg))
            fl = 4.0 * H * 64 * nseg * L * L
            res = {}
            for wpe in (2, 4):
                assert lib.rap_set_tuning(16, wpe) == 0

                def fx():
                    rc = lib.rap_x2_attention(_lib.ptr(qk), _lib.ptr(vt), nblk, _lib.ptr(cu), nseg, _lib.ptr(out), TP, H, _lib.ptr(ws), ws.numel(), st())
                    assert rc == 0, rc
                res[wpe] = timeit(fx)
            assert lib.rap_set_tuning(16, 2) == 0

            def ff():
                rc = lib.rap_attention_f32(_lib.ptr(qkv32), _lib.ptr(cu), nseg, _lib.ptr(out32), TP, H, _lib.ptr(bound), _lib.ptr(ws), ws.numel(), st())
                assert rc == 0, rc
            tf = timeit(ff, iters=2, warm=1)
            tx = min(res.values())
            emit({"kernel": f"attention[L={L}]", "x2_ms_one_block_per_cu": res[2] * 1e3, "x2_ms_two_blocks_per_cu": res[4] * 1e3, "f32_ms": tf * 1e3,
                  "x2_tflops_fp32_equiv": fl / tx / 1e12, "x2_frac_of_833TF": fl / tx / 1e12 / PEAK_X2, "f32_tflops": fl / tf / 1e12,
                  "f32_frac_of_157TF": fl / tf / 1e12 / PEAK_F32, "speedup": tf / tx})

    # ---- LayerNorm with paired output
    if args.only in ("", "ln"):
        x = torch.randn(TP, d, device=dev, generator=g)
        mod = torch.randn(1, 2 * d, device=dev, generator=g) * 0.1
        o = torch.zeros(TP, 2 * d, device=dev, dtype=torch.float16)

        def fl_():
            assert lib.rap_layernorm_mod_h16(3, _lib.ptr(x), _lib.ptr(o), TP, d, _lib.ptr(mod), 2 * d, None, st()) == 0
        t = timeit(fl_)
        emit({"kernel": "layernorm_x2", "ms": t * 1e3, "GB_per_s": TP * 4096 / t / 1e9, "bytes_per_token": 4096})


if __name__ == "__main__":
    main()
