#!/usr/bin/env python3
"""Driver of scripts/attn_mix.hip: prints one JSON line per (query blocks per wave, structure, waves per block, waves per SIMD)."""
import ctypes, json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libattn_mix.so")
src = os.path.join(here, "attn_mix.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", so])
if "--build-only" in sys.argv:
    sys.exit(0)
lib = ctypes.CDLL(so)
lib.attn_mix.restype = ctypes.c_double
lib.attn_mix.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_double)]
names = {0: "phase-serial + barrier (shipped structure)", 5: "phase-serial, no barrier", 1: "pipelined, 1 MFMA + 5 VALU per slot, pinned",
         2: "pipelined, row sums on v_pk_add_f32", 3: "pipelined, compiler's order", 4: "pipelined + barrier per tile", 6: "pipelined, compiler's order + barrier per tile", 7: "pipelined, asm slot blocks, not pinned", 8: "pipelined, compiler's order, P*V on the same quarter's P", 9: "pipelined, compiler's order, barrier per 4 tiles", 10: "pipelined, compiler's order, barrier per 2 tiles", 11: "as 8, barrier per 4 tiles", 12: "pipelined, compiler's order, barrier per 2 tiles + K/V stream (global -> registers -> LDS)", 13: "as 12, global loads only", 14: "as 12, ds_write only", 15: "as 12, all blocks stream the same 2 MB"}
CASES = [(1, 12, 8, 2), (1, 13, 8, 2), (1, 14, 8, 2), (1, 15, 8, 2), (1, 10, 8, 2), (1, 3, 8, 2)]
MORE = [(1, 8, 8, 2), (1, 9, 8, 2), (1, 10, 8, 2), (1, 11, 8, 2), (1, 8, 4, 3), (1, 9, 4, 3), (1, 9, 4, 2), (1, 11, 4, 3), (1, 3, 8, 2), (1, 6, 8, 2), (1, 0, 8, 4)]
ALL = [(1, 0, 8, 4), (1, 5, 8, 4), (1, 0, 4, 4), (1, 1, 8, 2), (1, 2, 8, 2), (1, 3, 8, 2), (1, 4, 8, 2), (1, 1, 4, 3), (1, 3, 4, 3),
         (1, 1, 4, 2), (1, 1, 4, 1), (2, 0, 4, 2), (2, 5, 4, 2), (2, 0, 8, 2), (2, 1, 4, 1), (2, 2, 4, 1), (2, 3, 4, 1), (2, 4, 4, 1),
         (2, 1, 4, 2), (2, 3, 4, 2), (2, 1, 8, 2),
         (1, 6, 8, 2), (1, 7, 8, 2), (1, 6, 4, 3), (1, 7, 4, 3), (1, 3, 4, 2), (1, 3, 4, 1), (2, 6, 4, 2), (2, 7, 4, 2)]
for qb, mode, waves, occ in (ALL if '--all' in sys.argv else CASES):
    per_cu = max(1, occ * 4 // waves)
    blocks = 256 * per_cu * 4
    tiles = 2048 // qb
    ms = ctypes.c_double()
    tf = lib.attn_mix(qb, mode, waves, occ, 0, blocks, tiles, ctypes.byref(ms))
    print(json.dumps({"query_blocks_per_wave": qb, "mode": mode, "what": names[mode], "waves_per_block": waves, "waves_per_simd": occ,
                      "blocks_per_cu": per_cu, "tflops": round(tf, 1), "frac_of_2500": round(tf / 2500.0, 3), "ms": round(ms.value, 3)}), flush=True)
