#!/usr/bin/env python3
"""Run scripts/ldsdma_mix.hip: L2-hit LDS-DMA streams with HBM-miss pieces mixed into every wave's queue vs confined to two waves."""
import ctypes, json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libldsdma_mix.so")
src = os.path.join(here, "ldsdma_mix.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", so])
if "--build-only" in sys.argv:
    sys.exit(0)
lib = ctypes.CDLL(so)
lib.ldsdma_mix.argtypes = [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_double)] * 3
for mode, every, what in ((0, 1, "all waves: shared L2-resident window"), (1, 8, "every wave: 1 of 8 pieces from a private HBM stream"),
                          (1, 4, "every wave: 1 of 4 pieces from HBM"), (1, 2, "every wave: 1 of 2 pieces from HBM"), (1, 1, "every wave: all pieces from HBM"),
                          (2, 1, "waves 0-1 HBM only, waves 2-7 L2 window only")):
    for rep in range(2):
        h, m, ms = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        rc = lib.ldsdma_mix(mode, every, 2048, ctypes.byref(h), ctypes.byref(m), ctypes.byref(ms))
        print(json.dumps({"what": what, "rc": rc, "GBps_per_cu_hit_class": round(h.value, 1), "GBps_per_cu_miss_class": round(m.value, 1),
                          "GBps_per_cu_total": round(h.value + m.value, 1) if mode != 2 else None, "ms": round(ms.value, 3)}), flush=True)
