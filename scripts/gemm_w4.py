#!/usr/bin/env python3
"""Run scripts/gemm_w4.hip (prototype: bf16 GEMM with four waves per 256 x 256 tile, one wave per SIMD, 128 x 128 per wave) on the
layer shapes and the guide's square shape.  One JSON line per (shape, mode): mode bit 0 = main loop only (no stores), bit 1 = persistent."""
import ctypes, json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
so, src = os.path.join(here, "libgemm_w4.so"), os.path.join(here, "gemm_w4.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", src, "-o", so])
if "--build-only" in sys.argv:
    sys.exit(0)
lib = ctypes.CDLL(so)
lib.gemm_w4.restype = ctypes.c_double
lib.gemm_w4.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_double)]
SHAPES = [("guide square", 8192, 8192, 8192), ("ff2", 262144, 512, 2048), ("out-projection", 262144, 512, 512), ("ff1 (no GEGLU)", 262144, 4096, 512),
          ("qkv (plain store)", 262144, 1536, 512)]
for name, M, N, K in SHAPES:
    modes = ((0, "one tile per block"), (2, "persistent"), (1, "one tile per block, main loop only"), (3, "persistent, main loop only"))
    if "--epilogue" in sys.argv:
        modes = ((18, "persistent, epilogue through an LDS slab (whole rows out)"), (16, "one tile per block, epilogue through an LDS slab"),
                 (2, "persistent, direct 8-byte stores"), (3, "persistent, main loop only"))
    if "--ablation" in sys.argv:
        modes = ((3, "persistent, main loop only"), (7, "main loop without LDS-DMA (MFMA + fragment reads + barrier)"),
                 (11, "main loop without fragment reads (MFMA + LDS-DMA + barrier)"), (15, "MFMA + barrier only"))
    if "--breg" in sys.argv:      # round 6: W packed in fragment order, global -> VGPR, the LDS-DMA stream carries A only
        modes = ((2, "persistent, both operands through LDS (direct 8-byte stores)"), (66, "persistent, W through REGISTERS (packed, global -> VGPR), A through LDS"),
                 (3, "persistent, main loop only, both operands through LDS"), (67, "persistent, main loop only, W through REGISTERS"))
    if "--zeros" in sys.argv:
        if name != "guide square":
            continue
        modes = ((15, "MFMA + barrier only, random operands"), (15 + 32, "MFMA + barrier only, ALL-ZERO operands"),
                 (3, "full main loop, random operands"), (3 + 32, "full main loop, ALL-ZERO operands"))
    for mode, what in modes:
        err = ctypes.c_double(-1.0)
        iters = 5 if M * N * K > 2 ** 38 else 20
        tf = lib.gemm_w4(M, N, K, mode, iters, ctypes.byref(err))
        print(json.dumps({"gemm": name, "M": M, "N": N, "K": K, "mode": what, "tflops": round(tf, 1), "frac_of_2500": round(tf / 2500, 3),
                          "max_rel_err_vs_naive": None if err.value < 0 else float(f"{err.value:.3g}")}), flush=True)
