"""ctypes binding of librapflow.so (the C ABI declared in include/rapflow.h).

The library is the product path.  If it is missing or cannot be loaded this module raises -- there
is NO CPU / PyTorch fallback anywhere in ``rap_amd`` (a silent fallback would void every parity and
performance claim).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librapflow.so")

ABI_VERSION = 6        # RAPFLOW_ABI_VERSION of include/rapflow.h this binding was written against
EPI_H_BIAS_RESID_H16 = 7   # rap_gemm_h16's fp16-residual epilogue (6 before ABI version 4; 6 is refused now)
# "float32x2" (round 5): split precision -- fp32-ACCURATE transformer blocks on the fp16 matrix pipe (include/rapflow.h, compute dtype 3)
DTYPES = {"float32": 0, "fp32": 0, "bfloat16": 1, "bf16": 1, "float16": 2, "fp16": 2, "float32x2": 3, "f32x2": 3}

ERRORS = {-1: "invalid argument", -2: "workspace too small", -3: "HIP runtime error", -4: "allocation failure"}


class RapError(RuntimeError):
    pass


class ModelDesc(ctypes.Structure):
    _fields_ = [("embed_dim", c_int32), ("num_layers", c_int32), ("num_heads", c_int32), ("local_feat_dim", c_int32)]


_P = c_void_p
# name -> (restype, argtypes); must list every symbol of include/rapflow.h (tests/test_abi.py checks).
SIGNATURES = {
    "rap_version": (c_int32, []),
    "rap_last_hip_error": (c_int32, []),
    "rap_weight_count": (c_int64, [ctypes.POINTER(ModelDesc)]),
    "rap_model_create": (c_int32, [ctypes.POINTER(ModelDesc), _P, c_int64, _P, ctypes.POINTER(_P)]),
    "rap_weight_count_latent": (c_int64, [ctypes.POINTER(ModelDesc), c_int32]),
    "rap_model_create_latent": (c_int32, [ctypes.POINTER(ModelDesc), c_int32, _P, c_int64, _P, ctypes.POINTER(_P)]),
    "rap_model_destroy": (None, [_P]),
    "rap_model_set_compute_dtype": (c_int32, [_P, c_int32, _P]),
    "rap_model_compute_dtype": (c_int32, [_P]),
    "rap_model_bounded_attention_launches": (c_int32, [_P]),
    "rap_model_set_qk_norm": (c_int32, [_P, c_int32]),
    "rap_model_qk_norm": (c_int32, [_P]),
    "rap_model_set_residual_dtype": (c_int32, [_P, c_int32]),
    "rap_model_residual_dtype": (c_int32, [_P]),
    "rap_workspace_bytes": (c_size_t, [_P, c_int64, c_int32, c_int32, c_int32]),
    "rap_dit_forward": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int64, _P, _P, _P, c_size_t, _P]),
    "rap_dit_forward_latent": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int64, _P, _P, _P, c_size_t, _P]),
    "rap_sample_latent": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int64, c_int32, c_int32, _P, _P, _P, _P,
                                    _P, _P, c_size_t, _P]),
    "rap_euler_step": (c_int32, [_P, _P, c_float, c_float, _P, _P, _P, c_int64, _P]),
    "rap_procrustes_workspace_bytes": (c_size_t, [c_int32]),
    "rap_fit_transformations": (c_int32, [_P, _P, _P, c_int32, c_int32, _P, _P, _P, c_size_t, _P]),
    "rap_rigidify": (c_int32, [_P, _P, _P, c_int32, c_int32, _P, _P, c_size_t, _P]),
    "rap_rigidify_blend": (c_int32, [_P, _P, _P, c_int32, c_int32, _P, c_float, c_float, _P, _P, c_size_t, _P]),
    "rap_sample": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int64, c_int32, c_int32, _P, _P, _P, _P,
                             _P, _P, c_size_t, _P]),
    "rap_rigidity_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "rap_rigidity_rmse": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32, _P, c_int32, _P, _P, c_size_t, _P]),
    "rap_trajectory_rigidity_rmse": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int64, c_int32, _P, _P, _P, _P, c_size_t, _P]),
    "rap_select_generation": (c_int32, [_P, c_int32, c_int32, c_int32, c_int64, _P, _P, _P, _P, c_int32, _P, _P, _P, _P, _P]),
    "rap_transform_errors": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, _P, _P, _P, _P, _P]),
    "rap_overlap_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "rap_overlap_ratio": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int64, _P, c_int32, _P, _P, _P, c_size_t, _P]),
    "rap_relative_transforms": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, _P, _P, _P, _P]),
    "rap_nn_metrics_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "rap_chamfer_rmse": (c_int32, [_P, _P, _P, c_int32, c_int64, _P, _P, c_size_t, _P]),
    "rap_correspondence_rmse": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_float, _P, _P, c_size_t, _P]),
    "rap_voxel_bounds": (c_int32, [_P, c_int64, c_float, _P, _P, _P]),
    "rap_voxel_table_slots": (c_int64, [_P]),
    "rap_voxel_workspace_bytes": (c_size_t, [_P]),
    "rap_voxel_coverage_workspace_bytes": (c_size_t, [_P]),
    "rap_voxel_coverage": (c_int32, [_P, c_int64, c_float, _P, _P, _P, c_size_t, _P]),
    "rap_voxel_downsample": (c_int32, [_P, c_int64, c_float, _P, c_float, _P, _P, _P, c_size_t, _P]),
    "rap_voxel_sorted_workspace_bytes": (c_size_t, [c_int64]),
    "rap_voxel_downsample_sorted": (c_int32, [_P, c_int64, c_float, _P, c_float, _P, _P, _P, c_size_t, _P]),
    "rap_voxel_coverage_sorted": (c_int32, [_P, c_int64, c_float, _P, _P, _P, c_size_t, _P]),
    "rap_farthest_point_sampling": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32, _P, _P, _P]),
    "rap_spinnet_weight_count": (c_int64, []),
    "rap_spinnet_create": (c_int32, [_P, c_int64, _P, ctypes.POINTER(_P)]),
    "rap_spinnet_destroy": (None, [_P]),
    "rap_spinnet_workspace_bytes": (c_size_t, [c_int32]),
    "rap_spinnet_describe": (c_int32, [_P, _P, _P, c_int64, _P, c_int32, c_float, c_int32, _P, c_int32, _P, c_size_t, _P]),
    "rap_outlier_workspace_bytes": (c_size_t, [c_int64]),
    "rap_statistical_outliers": (c_int32, [_P, c_int64, c_int32, ctypes.c_double, _P, _P, _P, _P, c_size_t, _P]),
    "rap_check_batch": (c_int32, [_P, _P, c_int32, c_int32, c_int64, _P, _P]),
    "rap_poison_on_flag": (c_int32, [_P, _P, c_int64, _P]),
    "rap_collate_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "rap_collate_transform": (c_int32, [_P, c_int32, _P, c_int32, c_int32, c_int64, _P, _P, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                        _P, _P, _P, c_size_t, _P]),
    "rap_gemm_f32": (c_int32, [c_int32, _P, c_int32, _P, c_int32, _P, c_int32, c_int32, c_int32, c_int32, _P, _P, c_int32,
                               _P, _P, c_int32, _P]),
    "rap_geglu_interleave": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, _P]),
    "rap_build_attention_worklist": (c_int32, [_P, c_int32, c_int32, _P, c_int32, _P, _P]),
    "rap_attention_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "rap_attention_f32": (c_int32, [_P, _P, c_int32, _P, c_int64, c_int32, _P, _P, c_size_t, _P]),
    "rap_layernorm_mod": (c_int32, [_P, _P, c_int64, c_int32, _P, c_int64, _P, _P]),
    "rap_layernorm_affine": (c_int32, [_P, _P, c_int64, c_int32, _P, _P, _P]),
    "rap_qknorm": (c_int32, [_P, c_int64, c_int32, _P, _P, _P]),
    "rap_posenc_x": (c_int32, [_P, _P, c_int64, _P]),
    "rap_posenc_static": (c_int32, [_P, _P, _P, _P, c_int32, _P, c_int64, _P]),
    "rap_token_sample": (c_int32, [_P, c_int32, _P, _P]),
    "rap_adaln_table": (c_int32, [_P, _P, c_int32, _P, _P, _P]),
    "rap_convert_h16": (c_int32, [c_int32, _P, _P, c_int64, _P]),
    "rap_gemm_h16": (c_int32, [c_int32, c_int32, _P, c_int32, _P, c_int32, _P, c_int32, c_int32, c_int32, c_int32, _P, _P,
                               c_int32, c_int32, _P, c_int32, _P]),
    "rap_gemm_h16_splitk_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "rap_gemm_h16_splitk": (c_int32, [c_int32, c_int32, _P, c_int32, _P, c_int32, _P, c_int32, c_int32, c_int32, c_int32, _P, _P, c_int32,
                                      _P, c_size_t, _P]),
    "rap_gemm_h16_qkvnorm": (c_int32, [c_int32, _P, c_int32, _P, c_int32, _P, c_int32, c_int32, c_int32, _P, _P, c_float, _P, c_int32, _P]),
    "rap_attention_h16": (c_int32, [c_int32, _P, _P, c_int32, _P, c_int32, _P, c_int64, c_int32, _P, _P, c_size_t, _P]),
    "rap_layernorm_mod_h16": (c_int32, [c_int32, _P, _P, c_int64, c_int32, _P, c_int64, _P, _P]),
    "rap_layernorm_affine_h16": (c_int32, [c_int32, _P, _P, c_int64, c_int32, _P, _P, _P]),
    "rap_qknorm_h16": (c_int32, [c_int32, _P, c_int64, c_int32, _P, _P, _P]),
    "rap_x2_pack": (c_int32, [_P, c_int64, c_int64, c_int32, c_float, _P, _P]),
    "rap_x2_unpack": (c_int32, [_P, c_int64, c_int32, c_float, _P, _P]),
    "rap_x2_gemm": (c_int32, [c_int32, _P, c_int32, _P, c_int32, _P, c_int32, c_int32, c_int32, c_int32, _P, _P, c_int32, c_float, c_int32,
                              _P, _P, c_float, _P, c_int32, _P]),
    "rap_x2_attention": (c_int32, [_P, _P, c_int32, _P, c_int32, _P, c_int64, c_int32, _P, c_size_t, _P]),
    "rap_set_tuning": (c_int32, [c_int32, c_int32]),
    "rap_profile_enable": (c_int32, [c_int32]),
    "rap_profile_reset": (c_int32, []),
    "rap_profile_collect": (c_int32, [_P, _P]),
    "rap_profile_collect_ex": (c_int32, [_P, _P, c_int32]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load librapflow.so or raise (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RapError(f"{LIB_PATH} not found: build it with `python -m rap_amd._build` "
                       "(or __graft_entry__.build()); rap_amd has no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.rap_version() != ABI_VERSION:
        raise RapError(f"{LIB_PATH} has ABI version {lib.rap_version()}, this binding expects {ABI_VERSION}: rebuild it "
                       "(python -m rap_amd._build --force)")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        lib = load()
        extra = f" (hipError {lib.rap_last_hip_error()})" if rc == -3 else ""
        raise RapError(f"{what} failed: {ERRORS.get(rc, rc)}{extra}")


def ptr(t) -> c_void_p:
    """Device pointer of a torch tensor (or NULL for None)."""
    return c_void_p(0) if t is None else c_void_p(t.data_ptr())


def current_stream(device) -> c_void_p:
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)
