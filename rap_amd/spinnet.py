"""Host-side mirror of the reference's MiniSpinNet local feature extractor.

``MiniSpinNet`` keeps the reference constructor / ``forward`` signature
(``dataset_process/utils/spinnet/patch_embedder.py:11-92``) and its ``state_dict`` contract, and owns no PyTorch math: the
descriptors come from ``rap_spinnet_describe`` in librapflow.  Only the configuration the reference ships is supported
(512 points per patch, 3 x 7 x 20 voxels, 10 samples per voxel, delta 0.8, global-z alignment; extract_sample_features.py:82-89,
demo.py:545-547).  ``forward`` returns ``{'desc': (K,32)}`` -- the only entry the reference's callers read
(extract_sample_features.py:203); the equivariant map / axis outputs of the training code are not produced.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from .flow_model import _f32c, _require_cuda, workspace

_CONV = [(16, 64, 27), (64, 64, 9), (64, 128, 9), (128, 128, 9), (128, 64, 9), (64, 64, 9), (64, 32, 9), (32, 32, 9)]


def spinnet_weight_spec():
    """(name, shape) of the float tensors of MiniSpinNet.state_dict() in registration order."""
    spec = [("pnt_layer.0.weight", (16, 3, 1, 1)), ("pnt_layer.0.bias", (16,)), ("pnt_layer.1.weight", (16,)),
            ("pnt_layer.1.bias", (16,)), ("pnt_layer.1.running_mean", (16,)), ("pnt_layer.1.running_var", (16,)),
            ("pool_layer.0.weight", (16, 32, 1, 1)), ("pool_layer.0.bias", (16,)), ("pool_layer.1.weight", (16,)),
            ("pool_layer.1.bias", (16,)), ("pool_layer.1.running_mean", (16,)), ("pool_layer.1.running_var", (16,)),
            ("pool_layer.3.weight", (1, 16, 1, 1)), ("pool_layer.3.bias", (1,)), ("pool_layer.4.weight", (1,)),
            ("pool_layer.4.bias", (1,)), ("pool_layer.4.running_mean", (1,)), ("pool_layer.4.running_var", (1,))]
    for i, (cin, cout, kk) in enumerate(_CONV):
        op = 3 * i
        shape = (cout, cin, 3, 3, 3) if kk == 27 else (cout, cin, 3, 3)
        spec += [(f"conv_net.ops.{op}.weight", shape), (f"conv_net.ops.{op}.bias", (cout,))]
        if i < 7:
            spec += [(f"conv_net.ops.{op + 1}.running_mean", (cout,)), (f"conv_net.ops.{op + 1}.running_var", (cout,))]
    return spec


def make_spinnet_weights(seed: int = 0) -> dict:
    """Seeded synthetic weights (no checkpoint offline): He-scaled convolutions, non-trivial BatchNorm statistics, and
    biases that keep the ReLUs of the attention pool active so that descriptors are not degenerate."""
    sd = {}
    for name, shape in spinnet_weight_spec():
        g = torch.Generator().manual_seed(seed * 7919 + sum(ord(c) for c in name))     # per tensor, by name
        if name.endswith("running_var"):
            t = torch.rand(shape, generator=g) * 0.8 + 0.4
        elif name.endswith("running_mean"):
            t = torch.randn(shape, generator=g) * 0.1
        elif name.endswith(".weight") and len(shape) == 1:
            t = torch.rand(shape, generator=g) * 0.8 + 0.6           # BatchNorm gain
        elif name.endswith(".weight"):
            fan_in = int(np.prod(shape[1:]))
            t = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
        else:
            t = torch.randn(shape, generator=g) * 0.05 + (0.3 if name.startswith("pool_layer") else 0.02)
        sd[name] = t.float()
    return sd


class MiniSpinNet:
    """Drop-in for ``dataset_process.utils.spinnet.MiniSpinNet`` (inference, descriptors only)."""

    def __init__(self, des_r: float = 3.0, num_points_per_patch: int = 512, rad_n: int = 3, azi_n: int = 20, ele_n: int = 7,
                 delta: float = 0.8, voxel_sample: int = 10, is_aligned_to_global_z: bool = True, keypoints_per_chunk: int = 2048):
        if (num_points_per_patch, rad_n, azi_n, ele_n, voxel_sample) != (512, 3, 20, 7, 10) or abs(delta - 0.8) > 1e-12:
            raise NotImplementedError("only the shipped configuration (512 pts, 3x7x20 voxels, 10 samples, delta 0.8) is built")
        self.is_aligned_to_global_z = bool(is_aligned_to_global_z)
        self.des_r, self.patch_sample = des_r, num_points_per_patch
        self.keypoints_per_chunk = int(keypoints_per_chunk)
        self._spec = spinnet_weight_spec()
        self._sd = None
        self._handle = ctypes.c_void_p(0)
        self._device = None

    def load_state_dict(self, state_dict: dict, strict: bool = True):
        """nn.Module contract: returns (missing_keys, unexpected_keys); strict=False keeps previous values of missing tensors."""
        from .flow_model import _IncompatibleKeys
        names = [n for n, _ in self._spec]
        missing = [n for n in names if n not in state_dict]
        unexpected = [k for k in state_dict if k not in set(names) and not k.endswith("num_batches_tracked")]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for MiniSpinNet: missing {missing[:4]}, unexpected {unexpected[:4]}")
        sd = dict(self._sd) if getattr(self, "_sd", None) else {}
        for n, shape in self._spec:
            if n not in state_dict:
                continue
            if tuple(state_dict[n].shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {n}: {tuple(state_dict[n].shape)} vs {shape}")
            sd[n] = state_dict[n].detach().to(torch.float32)
        self._sd = sd
        self._release()
        return _IncompatibleKeys(missing, unexpected)

    def eval(self):
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.RapError("rap_amd.MiniSpinNet runs on the GPU only (no CPU fallback)")
        self._ensure(device)
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def _release(self):
        if self._handle:
            _lib.load().rap_spinnet_destroy(self._handle)
            self._handle = ctypes.c_void_p(0)
            self._device = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _ensure(self, device):
        if self._handle and self._device == device:
            return
        if self._sd is None or any(n not in self._sd for n, _ in self._spec):
            raise _lib.RapError("load_state_dict() must be called before the model is used")
        self._release()
        lib = _lib.load()
        with torch.cuda.device(device):
            blob = torch.cat([self._sd[n].reshape(-1) for n, _ in self._spec]).to(device=device, dtype=torch.float32)
            assert blob.numel() == lib.rap_spinnet_weight_count()
            handle = ctypes.c_void_p(0)
            _lib.check(lib.rap_spinnet_create(_lib.ptr(blob), blob.numel(), _lib.current_stream(device), ctypes.byref(handle)),
                       "rap_spinnet_create")
            torch.cuda.current_stream(device).synchronize()
        self._handle, self._device = handle, device

    @torch.inference_mode()
    def forward(self, pts, kpts, des_r, is_aligned_to_global_z: bool = True, z_axis=None, is_aug: bool = False, perm=None):
        """pts (1,N,3), kpts (1,K,3) -> {'desc': (K,32)}.  ``perm`` (N,) overrides the shuffle the reference draws with
        ``np.random.choice(N, N, replace=False)`` (patch_embedder.py:99) -- by default the same call is made here, so the same
        numpy seed gives the same patches."""
        if z_axis is not None or is_aug:
            raise NotImplementedError("caller-supplied z axes and SO(2) augmentation are training-time options and are not built")
        _require_cuda(pts, "pts")
        device = pts.device
        self._ensure(device)
        p = _f32c(pts.reshape(-1, 3)); kp = _f32c(kpts.reshape(-1, 3).to(device))
        N, K = p.shape[0], kp.shape[0]
        if perm is None:
            perm = np.random.choice(N, N, replace=False)
        # the shuffle is applied once here (a gather: data movement only), so the kernel scans a contiguous cloud in the shuffled order
        p = p[torch.as_tensor(np.asarray(perm), dtype=torch.int64).to(device)].contiguous()
        perm_d = None
        desc = torch.empty((K, 32), dtype=torch.float32, device=device)
        lib = _lib.load()
        chunk = max(1, min(self.keypoints_per_chunk, K))
        ws = workspace(device, lib.rap_spinnet_workspace_bytes(chunk))
        flags = (0 if is_aligned_to_global_z else 1) | (2 if getattr(self, "im2col_path", False) else 0)     # rapflow.h: RAP_SPINNET_*
        with torch.cuda.device(device):
            rc = lib.rap_spinnet_describe(self._handle, _lib.ptr(p), _lib.ptr(perm_d), N, _lib.ptr(kp), K, float(des_r), flags, _lib.ptr(desc),
                                          chunk, _lib.ptr(ws), ws.numel(), _lib.current_stream(device))
        _lib.check(rc, "rap_spinnet_describe")
        return {"desc": desc}

    __call__ = forward
