#!/usr/bin/env python3
"""Per-block s_memtime stamps of the bf16 attention kernel (ablation build only: rap_debug_attn_ts + rap_set_tuning(3, 21)).
Stamps of thread 0: 0 entry, 1 work item + Q loads issued, 2 first K/V tile in LDS (barrier passed), 3 Q fragments arrived,
4 key loop done, 5 output stored."""
import ctypes, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rap_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.rap_debug_attn_ts.restype = ctypes.c_int
raw.rap_debug_attn_ts.argtypes = [ctypes.c_void_p]
assert lib.rap_set_tuning(3, 21) == 0
H, d = 8, 512
g = torch.Generator(device=dev).manual_seed(0)
for points, batch in ((1024, 128), (4096, 32)):
    TP = points * batch * 2
    nblk = TP // 64
    qk = (torch.nn.functional.normalize(torch.randn(2, H, TP, 64, device=dev, generator=g), dim=-1) * 8).to(torch.bfloat16)
    vt = torch.randn(H, nblk, 64, 64, device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty(TP, d, device=dev, dtype=torch.bfloat16)
    bound = torch.full((H,), 8.01, device=dev)
    cu = torch.arange(0, TP + 1, points, dtype=torch.int32, device=dev)
    nseg = cu.numel() - 1
    ws = torch.empty(lib.rap_attention_workspace_bytes(TP, nseg), dtype=torch.uint8, device=dev)
    nblocks = (TP // 256) * H
    ts = torch.zeros(nblocks * 8, dtype=torch.int64, device=dev)
    assert raw.rap_debug_attn_ts(ts.data_ptr()) == 0
    st = torch.cuda.current_stream(dev).cuda_stream
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.rap_attention_h16(1, _lib.ptr(qk), _lib.ptr(vt), nblk, _lib.ptr(cu), nseg, _lib.ptr(out), TP, H, _lib.ptr(bound),
                                   _lib.ptr(ws), ws.numel(), st)
        assert rc == 0
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    t = ts.cpu().numpy().reshape(nblocks, 8).astype(np.int64)
    r0 = t[:, 6].min()                       # s_memrealtime: 100 MHz, one counter for the whole device
    span_us = (t[:, 7].max() - r0) / 100.0
    dur_us = (t[:, 7] - t[:, 6]) / 100.0
    mhz = (t[:, 5] - t[:, 0]) / np.maximum(dur_us, 1e-3)
    seg = {"entry->item+Q issued": t[:, 1] - t[:, 0], "->first K/V tile in LDS": t[:, 2] - t[:, 1], "->Q arrived": t[:, 3] - t[:, 2],
           "key loop": t[:, 4] - t[:, 3], "normalise+store": t[:, 5] - t[:, 4], "whole block": t[:, 5] - t[:, 0]}
    row = {"L": points, "blocks": nblocks, "launch_ms_events": round(ms, 3), "first_entry_to_last_exit_us": round(float(span_us), 1),
           "block_duration_us": {"median": round(float(np.median(dur_us)), 1), "p10": round(float(np.percentile(dur_us, 10)), 1), "p90": round(float(np.percentile(dur_us, 90)), 1)},
           "shader_clock_MHz_during_blocks": {"median": round(float(np.median(mhz)), 0), "p10": round(float(np.percentile(mhz, 10)), 0), "p90": round(float(np.percentile(mhz, 90)), 0)},
           "sum_block_us_over_512_slots": round(float(dur_us.sum() / 512.0), 1)}
    for k, v in seg.items():
        row[k] = {"median": int(np.median(v)), "p10": int(np.percentile(v, 10)), "p90": int(np.percentile(v, 90)), "mean": round(float(v.mean()), 1)}
    start = (t[:, 6] - r0) / 100.0
    end = (t[:, 7] - r0) / 100.0
    grid = np.linspace(0, span_us, 42)[1:-1]
    row["blocks_alive_at_40_sample_times"] = [int(((start <= x) & (end > x)).sum()) for x in grid]
    order = np.sort(start)
    row["first_512_blocks_started_by_us"] = round(float(order[min(511, nblocks - 1)]), 1)
    print(json.dumps(row), flush=True)
