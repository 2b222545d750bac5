#!/usr/bin/env python3
"""Run scripts/ldsdma_fill.hip: global -> LDS DMA streaming rate per CU and chip-wide vs waves, pieces in flight, drain pattern and
source residency (see the header of the .hip file).  One JSON line per configuration; ~20 s on the GPU box."""
import ctypes, json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libldsdma_fill.so")
src = os.path.join(here, "ldsdma_fill.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", so])
if "--build-only" in sys.argv:
    sys.exit(0)
lib = ctypes.CDLL(so)
lib.ldsdma_fill.restype = ctypes.c_double
lib.ldsdma_fill.argtypes = [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_double)]
BLOCKS = 256
if "--shared" in sys.argv:
    # r02: one window per XCD, shared by its 32 CUs (the GEMM's weight operand): 1 MB (L2-resident) and 16 MB (beyond the 4 MB L2)
    for window_kb in (1024, 16384):
        for share, how in ((0, "private windows"), (1, "one window per XCD, lockstep"), (2, "one window per XCD, staggered starts")):
            lib.ldsdma_fill_set_share(share)
            for mode in (0, 1):
                for waves, depth in ((8, 4), (8, 8)):
                    per_iter = waves * depth * 1024
                    iters = max(64, (32 << 20) // per_iter)
                    ms = ctypes.c_double()
                    tbs = lib.ldsdma_fill(waves, depth, mode, window_kb, BLOCKS, iters, ctypes.byref(ms))
                    print(json.dumps({"window_kb": window_kb, "sharing": how, "mode": mode, "waves_per_cu": waves, "pieces_in_flight_per_wave": depth,
                                      "chip_TBps": round(tbs, 3), "GBps_per_cu": round(tbs * 1e3 / BLOCKS, 1), "ms": round(ms.value, 3)}), flush=True)
    sys.exit(0)
# 64 KB x 256 CUs = 16 MB: inside the 8 x 4 MB of L2; 256 KB x 256 = 64 MB: beyond L2, inside the 256 MB Infinity Cache
WINDOWS = ((64, "L2-resident window (64 KB per CU, re-read)"), (256, "Infinity-Cache-resident window (256 KB per CU, re-read)"),
           (65536, "HBM stream (64 MB per CU)"))
if "--l2-only" in sys.argv:
    WINDOWS = WINDOWS[:1]
for window_kb, where in WINDOWS:
    for mode, pattern in ((0, "drain each batch (vmcnt(0) + barrier)"), (1, "queue kept full (wait for the previous batch only)")):
        for waves in (1, 2, 4, 8):
            for depth in (1, 2, 4, 8):
                per_iter = waves * depth * 1024
                iters = max(64, (32 << 20) // per_iter)            # ~32 MB per CU per launch
                ms = ctypes.c_double()
                tbs = lib.ldsdma_fill(waves, depth, mode, window_kb, BLOCKS, iters, ctypes.byref(ms))
                if tbs < 0:
                    continue
                print(json.dumps({"source": where, "pattern": pattern, "waves_per_cu": waves, "pieces_in_flight_per_wave": depth,
                                  "kb_per_batch_per_cu": per_iter // 1024, "chip_TBps": round(tbs, 3),
                                  "GBps_per_cu": round(tbs * 1e3 / BLOCKS, 1), "ms": round(ms.value, 3)}), flush=True)
