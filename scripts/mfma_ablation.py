#!/usr/bin/env python3
import ctypes, json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libmfma_ablation.so")
src = os.path.join(here, "mfma_ablation.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", so])
if "--build-only" in sys.argv:
    sys.exit(0)
lib = ctypes.CDLL(so)
lib.ablate.restype = ctypes.c_double
lib.ablate.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
names = {0: "MFMA only", 1: "+ LDS operand reads", 2: "+ barrier / 64 MFMAs", 3: "+ global loads + ds_write (full loop)",
         4: "full loop without barrier"}
for blocks in (512, 2048):
    for mode in (0, 1, 2, 3, 4):
        ms = ctypes.c_double()
        tf = lib.ablate(mode, blocks, 4000 if blocks == 512 else 1000, ctypes.byref(ms))
        print(json.dumps({"blocks": blocks, "mode": mode, "what": names[mode], "tflops": round(tf, 1), "frac": round(tf / 157.3, 3),
                          "ms": round(ms.value, 2)}), flush=True)
