#!/usr/bin/env python3
"""Run scripts/mfma_peak.hip: fp32 MFMA ceiling with zero vs random operands, 1 and 2 waves per SIMD."""
import ctypes, json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libmfma_peak.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "mfma_peak.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-Wno-unused-value", "-Wno-unused-result", "-shared", "-fPIC",
                           os.path.join(here, "mfma_peak.hip"), "-o", so])
if "--build-only" in sys.argv:
    sys.exit(0)
lib = ctypes.CDLL(so)
lib.mfma_peak.restype = ctypes.c_double
lib.mfma_peak.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
for blocks, label in ((256, "1 wave/SIMD"), (512, "2 waves/SIMD"), (1024, "4 waves/SIMD")):
    for rnd in (0, 1):
        c, ms = ctypes.c_double(), ctypes.c_double()
        tf = lib.mfma_peak(blocks, 40000, rnd, ctypes.byref(c), ctypes.byref(ms))
        print(json.dumps({"bench": "v_mfma_f32_32x32x2_f32 only", "occupancy": label, "operands": "random" if rnd else "zeros",
                          "tflops": tf, "frac_of_157.3": tf / 157.3, "wall_ms": ms.value,
                          "implied_clock_GHz": tf / 157.3 * 2.4}), flush=True)
