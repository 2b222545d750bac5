"""Build librapflow.so (hipcc, gfx950 only) in-tree: rap_amd/csrc/*.hip -> rap_amd/csrc/librapflow.so.

Cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "librapflow.so")
SOURCES = ["api.hip", "gemm_f32.hip", "attn_f32.hip", "norm.hip", "embed.hip", "adaln.hip", "sampler_kernels.hip",
           "procrustes.hip", "gemm_h16.hip", "attn_h16.hip", "norm_h16.hip", "rigidity.hip", "transforms.hip", "overlap.hip", "spinnet.hip", "nn_metrics.hip", "fps.hip", "voxel.hip", "collate.hip", "outlier.hip", "voxel_sort.hip", "attn_x2.hip", "x2_pack.hip"]
HEADERS = ["common.h", "kernels.h", "kabsch.h", "half.h", os.path.join("..", "..", "include", "rapflow.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


ABLATION_LIB = os.path.join(CSRC, "librapflow_ablation.so")


def build(force: bool = False, verbose: bool = False, ablation: bool = False) -> str:
    """ablation=True: a SECOND library, librapflow_ablation.so, compiled with -DRAP_ABLATION_BUILD (kernel-variant switches and
    timestamp hooks for scripts/; never loaded by rap_amd itself)."""
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    flags = FLAGS + (["-DRAP_ABLATION_BUILD"] if ablation else [])
    lib = ABLATION_LIB if ablation else LIB
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".abl.o" if ablation else ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc, *flags, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, ablation="--ablation" in sys.argv))
