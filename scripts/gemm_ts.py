#!/usr/bin/env python3
"""Where a 256 x 256 tile of the phase-split 16-bit GEMM spends its time (ablation library: python -m rap_amd._build --ablation).
s_memrealtime stamps (100 MHz) of thread 0 of every block: entry, prologue DMA issued, first k-tile landed, k-loop done, epilogue done
(stores issued); blocks are grouped by the CU they ran on (XCC_ID, HW_ID) to get the gap between a block's last stamp and the entry
of the next block on the same CU (store drain + dispatch).  JSON lines."""
import ctypes, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rap_amd import _build

dev = torch.device("cuda:0")
raw = ctypes.CDLL(_build.ABLATION_LIB)
raw.rap_debug_gemm_ts.restype = ctypes.c_int
raw.rap_debug_gemm_ts.argtypes = [ctypes.c_void_p]
P = ctypes.c_void_p
raw.rap_gemm_h16.restype = ctypes.c_int
raw.rap_gemm_h16.argtypes = [ctypes.c_int32, ctypes.c_int32, P, ctypes.c_int32, P, ctypes.c_int32, P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                             ctypes.c_int32, P, P, ctypes.c_int32, ctypes.c_int32, P, ctypes.c_int32, P]
g = torch.Generator(device=dev).manual_seed(0)
TP = 262144
st = lambda: ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)  # noqa: E731
ptr = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())          # noqa: E731
PERSISTENT = "--persistent" in sys.argv
raw.rap_set_tuning.restype = ctypes.c_int; raw.rap_set_tuning.argtypes = [ctypes.c_int32, ctypes.c_int32]
assert raw.rap_set_tuning(11, 1 if PERSISTENT else 0) == 0
cases = [("plain 16-bit out", 0, 512, 512), ("plain 16-bit out", 0, 512, 2048), ("plain 16-bit out", 0, 1536, 512), ("GEGLU", 3, 4096, 512),
         ("fp16 residual", 7, 512, 512), ("qkv + qk-norm", 5, 1536, 512)]
raw.rap_gemm_h16_qkvnorm.restype = ctypes.c_int
raw.rap_gemm_h16_qkvnorm.argtypes = [ctypes.c_int32, P, ctypes.c_int32, P, ctypes.c_int32, P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, P, P, ctypes.c_float, P,
                                     ctypes.c_int32, P]


def xcd_remap(bid, nblk):
    q, r = nblk >> 3, nblk & 7
    xcd, idx = bid & 7, bid >> 3
    base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
    return base + idx
for name, epi, N, K in cases:
    A = torch.randn(TP, K, device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    Cw = N // 2 if epi == 3 else N
    C = torch.zeros(TP, Cw, device=dev, dtype=torch.float16 if epi == 7 else torch.bfloat16)
    resid = C if epi == 7 else None
    vt = torch.zeros(8 * (TP // 64) * 64 * 64, device=dev, dtype=torch.bfloat16) if epi == 5 else None
    gq = torch.ones(8, 64, device=dev)
    nblocks = (TP // 256) * (N // 256)
    ts = torch.zeros(max(nblocks * 8, 256 * 32 * 4), dtype=torch.int64, device=dev)
    assert raw.rap_debug_gemm_ts(ts.data_ptr()) == 0
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if epi == 5:
            rc = raw.rap_gemm_h16_qkvnorm(1, ptr(A), K, ptr(W), K, ptr(C), TP, K, 8, ptr(gq), ptr(gq), 8.0, ptr(vt), TP // 64, st())
        else:
            rc = raw.rap_gemm_h16(1, epi, ptr(A), K, ptr(W), K, ptr(C), Cw, TP, N, K, ptr(bias), ptr(resid), Cw if resid is not None else 0, 0,
                                  ptr(None), 0, st())
        assert rc == 0, rc
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    if PERSISTENT:
        # [block][tile][4]: k-loop start, k-loop end, epilogue end; consecutive tiles of one block give the tile-to-tile period
        tt = ts.cpu().numpy()[: 256 * 32 * 4].reshape(256, 32, 4).astype(np.int64)
        ntile = min(32, nblocks // 256)
        tt = tt[:, :ntile]
        us = lambda a: a / 100.0   # noqa: E731
        row = {"case": name, "N": N, "K": K, "persistent": True, "tiles_per_block": nblocks / 256, "launch_ms_events": round(ms, 3)}
        seg = {"k-loop": us(tt[:, :, 1] - tt[:, :, 0]).ravel(), "epilogue (to stores issued)": us(tt[:, :, 2] - tt[:, :, 1]).ravel(),
               "tile period (k-loop start -> next k-loop start)": us(tt[:, 1:, 0] - tt[:, :-1, 0]).ravel(),
               "epilogue end -> next k-loop start": us(tt[:, 1:, 0] - tt[:, :-1, 2]).ravel(),
               "block entry (first k-loop start) spread": us(tt[:, 0, 0] - tt[:, 0, 0].min())}
        for k, v in seg.items():
            row[k] = {"median": round(float(np.median(v)), 2), "p10": round(float(np.percentile(v, 10)), 2), "p90": round(float(np.percentile(v, 90)), 2)}
        if epi == 5:       # q / k column tiles (swapped product, direct 16-byte row stores) vs V tiles (V^T image through the LDS slab)
            nt_ = N // 256
            ncol = np.array([[xcd_remap(b + 256 * i, nblocks) % nt_ for i in range(ntile)] for b in range(256)])
            ep = us(tt[:, :, 2] - tt[:, :, 1])
            for lab, mask in (("epilogue, q / k tiles (swapped, direct stores)", ncol < 4), ("epilogue, V tiles (V^T through LDS)", ncol >= 4)):
                v = ep[mask]
                row[lab] = {"median": round(float(np.median(v)), 2), "p10": round(float(np.percentile(v, 10)), 2), "p90": round(float(np.percentile(v, 90)), 2)}
        print(json.dumps(row), flush=True)
        continue
    t = ts.cpu().numpy().reshape(-1, 8)[:nblocks].astype(np.int64)
    r0 = t[:, 0].min()
    us = lambda a: a / 100.0   # noqa: E731
    seg = {"entry->prologue DMA issued": us(t[:, 1] - t[:, 0]), "->first k-tile landed": us(t[:, 2] - t[:, 1]), "k-loop": us(t[:, 3] - t[:, 2]),
           "epilogue (to stores issued)": us(t[:, 4] - t[:, 3]), "whole block (thread 0)": us(t[:, 4] - t[:, 0])}
    # gap to the next block on the same CU
    cu = t[:, 5] & ~0xff & 0xffffffff | ((t[:, 5] >> 16) << 32)      # drop wave / SIMD / pipe ids: one key per CU (XCC, SE, SH, CU)
    gaps = []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        o = idx[np.argsort(t[idx, 0])]
        gaps.extend(us(t[o[1:], 0] - t[o[:-1], 4]))
    gaps = np.array(gaps)
    row = {"case": name, "N": N, "K": K, "blocks": nblocks, "cus": int(len(np.unique(cu))), "launch_ms_events": round(ms, 3),
           "first_entry_to_last_stamp_us": round(float(us(t[:, 4].max() - r0)), 1), "rounds": round(nblocks / len(np.unique(cu)), 2)}
    for k, v in seg.items():
        row[k] = {"median": round(float(np.median(v)), 2), "p10": round(float(np.percentile(v, 10)), 2), "p90": round(float(np.percentile(v, 90)), 2)}
    row["gap: stores issued -> next block's entry on that CU"] = {"median": round(float(np.median(gaps)), 2), "p10": round(float(np.percentile(gaps, 10)), 2),
                                                                   "p90": round(float(np.percentile(gaps, 90)), 2)}
    print(json.dumps(row), flush=True)
