#!/usr/bin/env python3
"""Per-kernel micro-benchmark at the BASELINE configs[1] shapes (TP = 32 pairs x 2 x 4096 = 262144 tokens).

Times every kernel class of the hot path through the C ABI with HIP events on the launch stream and prints
achieved TFLOP/s (MFMA-bound kernels, vs the 157.3 TF fp32 matrix peak) or GB/s (HBM-bound kernels, vs 8 TB/s
spec / 6.3 TB/s achievable) from the ALGORITHMIC bytes/flops stated in DESIGN.md.  Writes JSON lines.
"""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rap_amd import _lib  # noqa: E402
from rap_amd.flow_model import workspace  # noqa: E402


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


def bench_h16(args, lib, dev, st, TP, d, H, g):
    """16-bit MFMA twins at the same shapes; peak = 2.5 PFLOP/s dense bf16/fp16 (MI355X_MICROARCH.md)."""
    dt = _lib.DTYPES[args.dtype]
    tdt = {1: torch.bfloat16, 2: torch.float16}[dt]
    PEAK = 2500.0
    rows = []
    nblk = (TP + 255) // 256 * 256 // 64

    def gemm_case(name, epi, N, K, out_half, Cw=None, heads=0):
        lda = K + args.pad_lda                     # r02: padded row strides (is the k-tile fetch camping on a few L2 channels?)
        A = torch.randn(TP, lda, device=dev, generator=g).to(tdt)
        W = (torch.randn(N, lda, device=dev, generator=g) / K ** 0.5).to(tdt)
        bias = torch.randn(N, device=dev, generator=g)
        Cw = Cw or N
        C = torch.zeros(TP * Cw, device=dev, dtype=tdt if out_half else torch.float32)
        vt = torch.zeros(H * nblk * 64 * 64, device=dev, dtype=tdt) if epi == 4 else None
        resid = C if epi == 1 else None
        def fn():
            rc = lib.rap_gemm_h16(dt, epi, _lib.ptr(A), lda, _lib.ptr(W), lda, _lib.ptr(C), Cw, TP, N, K, _lib.ptr(bias),
                                  _lib.ptr(resid), Cw if epi == 1 else 0, heads, _lib.ptr(vt), nblk if epi == 4 else 0, st())
            assert rc == 0, rc
        t = timeit(fn)
        fl = 2.0 * TP * N * K
        rows.append({"kernel": f"gemm_h16[{name}]", "dtype": args.dtype, "pad_lda": args.pad_lda, "M": TP, "N": N, "K": K,
                     "ms": t * 1e3, "tflops": fl / t / 1e12, "frac_of_2500TF": fl / t / 1e12 / PEAK})

    if args.only in ("", "gemm"):
        gemm_case("qkv split + V^T", 4, 3 * d, d, True, Cw=2 * d, heads=H)
        gemm_case("out_proj +bias +resid (fp32 out)", 1, d, d, False)
        gemm_case("ff1 GEGLU", 3, 8 * d, d, True, Cw=4 * d)
        gemm_case("ff2 +bias +resid (fp32 out)", 1, d, 4 * d, False)
    if args.only in ("", "attention"):
        qk = torch.nn.functional.normalize(torch.randn(2, H, TP, 64, device=dev, generator=g), dim=-1) * 8
        qk = qk.to(tdt)
        vt = torch.randn(H, nblk, 64, 64, device=dev, generator=g).to(tdt)
        out = torch.empty(TP, d, device=dev, dtype=tdt)
        for bounded, (name, L) in [(b, c) for b in ((1, 0) if args.bounded == 2 else (args.bounded,))
                                   for c in (("per part", args.points), ("per sample", args.points * args.views))]:
            bound = torch.full((H,), 8.01, device=dev) if bounded else None     # |q| = |k| = 8  ->  q.k/8 <= 8
            cu = torch.arange(0, TP + 1, L, dtype=torch.int32, device=dev)
            nseg = cu.numel() - 1
            ws = workspace(dev, lib.rap_attention_workspace_bytes(TP, nseg))
            def fn():
                rc = lib.rap_attention_h16(dt, _lib.ptr(qk), _lib.ptr(vt), nblk, _lib.ptr(cu), nseg, _lib.ptr(out), TP, H,
                                           _lib.ptr(bound), _lib.ptr(ws), ws.numel(), st())
                assert rc == 0, rc
            t = timeit(fn, iters=1 if args.pmc else 5, warm=0 if args.pmc else 2)
            fl = 4.0 * H * 64 * L * TP
            rows.append({"kernel": f"attention_h16[{name} L={L}]", "dtype": args.dtype, "bounded": bool(bounded), "ms": t * 1e3, "tflops": fl / t / 1e12,
                         "frac_of_2500TF": fl / t / 1e12 / PEAK})
    if args.only == "":
        x = torch.randn(TP, d, device=dev, generator=g); y = torch.empty(TP, d, device=dev, dtype=tdt)
        mod = torch.randn(2 * d, device=dev, generator=g)
        def mem_case(name, fn, nbytes):
            t = timeit(fn, iters=10, warm=2)
            rows.append({"kernel": name, "dtype": args.dtype, "ms": t * 1e3, "algorithmic_GB": nbytes / 1e9, "GBps": nbytes / t / 1e9,
                         "frac_of_8TBps": nbytes / t / 8e12})
        mem_case("layernorm_mod_h16 (fp32 in, 16-bit out)",
                 lambda: lib.rap_layernorm_mod_h16(dt, _lib.ptr(x), _lib.ptr(y), TP, d, _lib.ptr(mod), 0, _lib.ptr(None), st()), TP * d * 6)
        qk2 = torch.randn(2, H, TP, 64, device=dev, generator=g).to(tdt)
        gq = torch.ones(H, 64, device=dev)
        mem_case("qknorm_h16 (q,k in place)", lambda: lib.rap_qknorm_h16(dt, _lib.ptr(qk2), TP, H, _lib.ptr(gq), _lib.ptr(gq), st()),
                 2 * TP * d * 2 * 2)
    for r in rows:
        print(json.dumps(r))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--only", default="", help="'attention' or 'gemm': restrict to one kernel family (PMC passes)")
    ap.add_argument("--pmc", action="store_true", help="one launch per kernel, no warm-up (for rocprofv3 --pmc passes)")
    ap.add_argument("--dtype", default="float32", help="float32 | bfloat16 | float16 (16-bit: GEMM / attention / LN / qknorm twins)")
    ap.add_argument("--tuning", action="append", default=[], metavar="KEY=VALUE", help="rap_set_tuning(KEY, VALUE) before the run")
    ap.add_argument("--pad-lda", type=int, default=0, help="16-bit GEMM: extra elements per row of A and W (row stride K + pad)")
    ap.add_argument("--bounded", type=int, default=1, help="attention: 1 = pass per-head logit bounds (bounded softmax kernel), 0 = none (online "
                                                            "softmax kernel), 2 = time both")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    for kv in args.tuning:
        k_, v_ = (int(x) for x in kv.split('='))
        assert lib.rap_set_tuning(k_, v_) == 0, kv
    st = lambda: _lib.current_stream(dev)
    TP = args.batch * args.views * args.points
    d, H = 512, 8
    g = torch.Generator(device=dev).manual_seed(0)
    rows = []
    if args.dtype != "float32":
        return bench_h16(args, lib, dev, st, TP, d, H, g)

    def gemm_case(name, epi, N, K, ldc=None, heads=0):
        A = torch.randn(TP, K, device=dev, generator=g)
        W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
        bias = torch.randn(N, device=dev, generator=g)
        Cw = (N // 2) if epi == 3 else N
        C = torch.zeros(TP * Cw, device=dev)
        resid = C.view(TP, Cw) if epi == 1 else None
        def fn():
            rc = lib.rap_gemm_f32(epi, _lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(C), ldc or Cw, TP, N, K, _lib.ptr(bias),
                                  _lib.ptr(resid), Cw if epi == 1 else 0, _lib.ptr(None), _lib.ptr(None), heads, st())
            assert rc == 0, rc
        t = timeit(fn)
        fl = 2.0 * TP * N * K
        rows.append({"kernel": f"gemm_f32[{name}]", "M": TP, "N": N, "K": K, "ms": t * 1e3, "tflops": fl / t / 1e12,
                     "frac_of_157.3TF": fl / t / 1e12 / 157.3})

    if args.only in ("", "gemm"):
        gemm_case("qkv headmajor", 4, 3 * d, d, heads=H)
        gemm_case("out_proj +bias +resid", 1, d, d)
        gemm_case("ff1 GEGLU", 3, 8 * d, d)
        gemm_case("ff2 +bias +resid", 1, d, 4 * d)
        gemm_case("embed K=64 +resid", 1, d, 64)
        gemm_case("head silu 512->256", 2, d // 2, d)
    if args.only == "gemm":
        for r in rows:
            print(json.dumps(r))
        return

    # attention
    qkv = torch.randn(3, H, TP, 64, device=dev, generator=g)
    qkv[:2] = torch.nn.functional.normalize(qkv[:2], dim=-1) * 8
    out = torch.empty(TP, d, device=dev)
    for bounded in ((1, 0) if args.bounded == 2 else (args.bounded,)):
        bound32 = torch.full((H,), 8.01, device=dev) if bounded else None     # |q| = |k| = 8  ->  q.k/8 <= 8
        for name, L in (("per part", args.points), ("per sample", args.points * args.views)):
            cu = torch.arange(0, TP + 1, L, dtype=torch.int32, device=dev)
            nseg = cu.numel() - 1
            ws = workspace(dev, lib.rap_attention_workspace_bytes(TP, nseg))
            def fn():
                rc = lib.rap_attention_f32(_lib.ptr(qkv), _lib.ptr(cu), nseg, _lib.ptr(out), TP, H, _lib.ptr(bound32), _lib.ptr(ws), ws.numel(), st())
                assert rc == 0, rc
            t = timeit(fn, iters=1 if args.pmc else 3, warm=0 if args.pmc else 1)
            fl = 4.0 * H * 64 * L * TP
            rows.append({"kernel": f"attention_f32[{name} L={L}]", "bounded": bool(bounded), "ms": t * 1e3, "tflops": fl / t / 1e12,
                         "frac_of_157.3TF": fl / t / 1e12 / 157.3})

    if args.only == "attention":
        for r in rows:
            print(json.dumps(r))
        return

    # HBM-bound ring
    def mem_case(name, fn, nbytes):
        t = timeit(fn, iters=10, warm=2)
        rows.append({"kernel": name, "ms": t * 1e3, "algorithmic_GB": nbytes / 1e9, "GBps": nbytes / t / 1e9,
                     "frac_of_8TBps": nbytes / t / 8e12})

    x = torch.randn(TP, d, device=dev, generator=g); y = torch.empty_like(x)
    mod = torch.randn(2 * d, device=dev, generator=g)
    mem_case("layernorm_mod (adaLN)", lambda: lib.rap_layernorm_mod(_lib.ptr(x), _lib.ptr(y), TP, d, _lib.ptr(mod), 0, _lib.ptr(None), st()),
             TP * d * 4 * 2)
    gq = torch.ones(H, 64, device=dev)
    mem_case("qknorm (q,k in place)", lambda: lib.rap_qknorm(_lib.ptr(qkv), TP, H, _lib.ptr(gq), _lib.ptr(gq), st()), 2 * TP * d * 4 * 2)
    x3 = torch.randn(TP, 3, device=dev, generator=g); ax = torch.empty(TP, 64, device=dev)
    mem_case("posenc_x", lambda: lib.rap_posenc_x(_lib.ptr(x3), _lib.ptr(ax), TP, st()), TP * (12 + 256))
    v3 = torch.randn(TP, 3, device=dev, generator=g); x0 = torch.empty_like(x3); xn = torch.empty_like(x3); tr = torch.empty_like(x3)
    mem_case("euler_step", lambda: lib.rap_euler_step(_lib.ptr(x3), _lib.ptr(v3), 0.5, 0.05, _lib.ptr(x0), _lib.ptr(xn), _lib.ptr(tr), TP * 3, st()),
             TP * 60)
    ppp = torch.full((args.batch, args.views), args.points, dtype=torch.int64, device=dev)
    R = torch.empty(args.batch, args.views, 3, 3, device=dev); tt = torch.empty(args.batch, args.views, 3, device=dev)
    wsp = workspace(dev, lib.rap_procrustes_workspace_bytes(args.batch * args.views))
    mem_case("procrustes fit (moments+solve)", lambda: lib.rap_fit_transformations(_lib.ptr(x3), _lib.ptr(v3), _lib.ptr(ppp), args.batch, args.views,
                                                                                 _lib.ptr(R), _lib.ptr(tt), _lib.ptr(wsp), wsp.numel(), st()), TP * 24)
    mem_case("rigidify+blend (fit + apply)", lambda: lib.rap_rigidify_blend(_lib.ptr(v3), _lib.ptr(x3), _lib.ptr(ppp), args.batch, args.views, _lib.ptr(x0),
                                                                            0.6, 0.4, _lib.ptr(xn), _lib.ptr(wsp), wsp.numel(), st()), TP * (24 + 36))
    # caller-side rows: generation selection criteria on a 20-step trajectory of the same batch
    import rap_amd
    cond = torch.rand(TP, 3, device=dev, generator=g) - 0.5
    traj = cond[None] + 0.01 * torch.randn(20, TP, 3, device=dev, generator=g)
    cu = torch.arange(0, TP + 1, args.views * args.points, dtype=torch.int32, device=dev)
    sc = torch.ones(args.batch, device=dev)
    mem_case("trajectory rigidity RMSE (20 x [Procrustes fit + RMSE])",
             lambda: rap_amd.average_trajectory_rigidity_rmse(cond, traj, ppp, cu, sc), 20 * TP * 48)
    t_ov = timeit(lambda: rap_amd.compute_overlap_ratio(cond, ppp, cu, [0.005, 0.01, 0.02]), iters=5, warm=1)
    pairs = args.batch * float(args.views * args.points) ** 2
    rows.append({"kernel": "overlap ratio (cross-part nearest neighbour, N^2 in LDS)", "ms": t_ov * 1e3, "pair_distances": pairs,
                 "Gpairs_per_s": pairs / t_ov / 1e9})
    # the step before the path: MiniSpinNet descriptors for 4096 keypoints of a 65536-point cloud
    from rap_amd.spinnet import MiniSpinNet, make_spinnet_weights
    net = MiniSpinNet(des_r=0.2); net.load_state_dict(make_spinnet_weights(0)); net.to(dev)
    cloud = torch.rand(65536, 3, device=dev, generator=g) * torch.tensor([4.0, 4.0, 0.3], device=dev)
    kp = cloud[:4096].clone()
    import numpy as np
    perm = np.random.RandomState(0).permutation(65536)
    t_sp = timeit(lambda: net(cloud[None], kp[None], 0.2, True, perm=perm), iters=3, warm=1)
    rows.append({"kernel": "MiniSpinNet descriptors (65536 pts, 4096 keypoints)", "ms": t_sp * 1e3, "keypoints_per_s": 4096 / t_sp,
                 "algorithmic_TFLOPs": 4096 * 119e6 / t_sp / 1e12})
    # preprocessing in front of it: voxel down-sampling of a 2M-point scan (60 x 60 x 8 m at 0.2 m voxels: a 27M-slot key table)
    from rap_amd.point_sampling import voxel_down_sample_torch, sample_farthest_points
    scan = (torch.rand(2_000_000, 3, device=dev, generator=g) - 0.5) * torch.tensor([60.0, 60.0, 8.0], device=dev)
    t_vx = timeit(lambda: voxel_down_sample_torch(scan, 0.2), iters=5, warm=1)
    kept = voxel_down_sample_torch(scan, 0.2).numel()
    rows.append({"kernel": "voxel down-sampling (2M points, 0.2 m voxels, incl. 2 host syncs)", "ms": t_vx * 1e3, "kept": kept,
                 "Mpoints_per_s": 2.0 / t_vx})
    fps_in = scan[:64 * 16384].reshape(64, 16384, 3).contiguous()
    t_fps = timeit(lambda: sample_farthest_points(fps_in, K=2048), iters=3, warm=1)
    rows.append({"kernel": "farthest point sampling (64 clouds x 16384 points -> 2048 each)", "ms": t_fps * 1e3,
                 "picks_per_s": 64 * 2048 / t_fps})
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
