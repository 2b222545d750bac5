"""Host-side mirror of the reference's velocity network class.

``PointCloudDiT`` keeps the reference constructor and ``forward`` signature
(reference ``rectified_point_flow/flow_model/point_cloud_dit.py:20-36,141-191``) and the reference
``state_dict`` key/shape contract, but owns no PyTorch math: ``forward``
hands device pointers to ``rap_dit_forward`` in librapflow (hand-written gfx950 kernels).
PyTorch is used for device memory and the current stream only.
"""
from __future__ import annotations

import collections
import os
import ctypes

import torch

from . import _lib
from .synthetic import weight_spec

_WORKSPACES: dict = {}


def workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Grow-only scratch buffer handed to the C ABI as the caller-owned workspace: one per (device, current stream) -- calls that
    are in flight on different streams (RectifiedPointFlow's concurrent batch shards) must not share scratch."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (device.type, idx, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    buf = _WORKSPACES.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _WORKSPACES.pop(key, None)
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _WORKSPACES[key] = buf
    return buf


def _require_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise _lib.RapError(f"{name} must live on the GPU: rap_amd has no CPU path (got device {t.device})")


# "auto": fp16 residual stream under bf16 compute, fp32 under fp16 compute -- chosen by measurement against the reference's fp32
# fixtures (r03 call 4, tests/test_headline_gpu.py): with bf16 operands the fp16 stream does not move the deviation (configs[1] final
# cloud 4.8e-4 vs 5.4e-4, |dR|_F 8.9e-4 vs 1.2e-3; configs[3] 1.8e-3 vs 1.2e-3) and buys 2.4 % of a sampling call; with fp16 operands
# it doubles to triples it (2.3e-4 vs 1.1e-4, 6.9e-4 vs 2.2e-4), so that mode keeps the fp32 stream.  DESIGN.md section 4.3.
DEFAULT_RESIDUAL_DTYPE = "auto"


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.float32).contiguous()


_IncompatibleKeys = collections.namedtuple("IncompatibleKeys", ["missing_keys", "unexpected_keys"])   # torch.nn.Module's return type


class PointCloudDiT:
    """Drop-in for ``rectified_point_flow.flow_model.PointCloudDiT`` (inference only)."""

    def __init__(self, in_dim: int, out_dim: int, embed_dim: int, num_layers: int, num_heads: int,
                 dropout_rate: float = 0.0, softcap: float = 0.0, qk_norm: bool = True, attn_dtype: str = "float16",
                 final_mlp_act=None, max_points_per_part: int = 500, max_points_per_batch: int = 40000,
                 scale_emb_on: bool = True, local_feat_concat_on: bool = True, local_feat_dim: int = 0,
                 compute_dtype: str | None = "float32", residual_dtype: str | None = None):
        # in_dim > 0 (round 6; embedding.py:107-118,163-166; modeling.py:788 builds in_dim = 64): `latent_features` (TP, in_dim) are
        # concatenated into the embedding input between the coordinate and the scale embeddings.  Off in every shipped config
        # (encoder_on: false) but part of the cited files: the native model takes them as in_dim more columns of the hoisted embedding GEMM.
        if in_dim < 0 or in_dim % 4 != 0 or in_dim > 512:
            raise NotImplementedError("in_dim must be a multiple of 4 in [0, 512]")
        if out_dim != 3:
            raise NotImplementedError("out_dim must be 3")
        if dropout_rate != 0.0 or softcap != 0.0:
            raise NotImplementedError("dropout / softcap are 0 in inference (layer.py:28-29)")
        # qk_norm / scale_emb_on / local_feat_concat_on (point_cloud_dit.py:28,33-34): every shipped configuration leaves them True; since
        # round 5 the other settings are honoured too (VERDICT r04 missing 3).  The two embedding switches only narrow emb_proj's input
        # (embedding.py:116-118,169-177): the native embedding GEMM runs on the full-width packing with ZERO weight columns for an absent
        # input -- exact, adding 0.0 changes no partial sum.  qk_norm=False skips MultiHeadRMSNorm (layer.py:75-83,103-104): the attention
        # launches then take the online-softmax kernels (no norm, no logit bound).
        self.qk_norm, self.scale_emb_on, self.local_feat_concat_on = bool(qk_norm), bool(scale_emb_on), bool(local_feat_concat_on)
        if attn_dtype not in ("float16", "fp16", "bfloat16", "bf16", "float32", "fp32", "float32x2", "f32x2"):      # (the last two: this package's split-precision mode)
            raise ValueError(f"Unsupported attn_dtype: {attn_dtype}")   # point_cloud_dit.py:80-81
        if embed_dim != 64 * num_heads:
            raise NotImplementedError("head_dim must be 64")
        self.in_dim, self.out_dim, self.embed_dim = in_dim, out_dim, embed_dim
        self.num_layers, self.num_heads, self.local_feat_dim = num_layers, num_heads, local_feat_dim
        self.max_points_per_part, self.max_points_per_batch = max_points_per_part, max_points_per_batch
        # attn_dtype is kept for interface parity.  What selects the arithmetic of the transformer blocks is
        # compute_dtype (an extension): "float32" (default) = exact-fp32 MFMA everywhere, i.e. at or above any
        # attn_dtype the reference accepts; "bfloat16" / "float16" = 16-bit MFMA GEMMs + attention with fp32
        # accumulation (what the reference's GPU inference runs under Lightning "16-mixed" autocast,
        # trainer/infer.yaml:6); "float32x2" = SPLIT PRECISION (round 5): fp32-accurate blocks on the fp16 matrix pipe -- operands as
        # fp16 head + tail, three products per contraction, results at "float32"'s distance from fp64 at several times its speed;
        # None = follow torch autocast at call time, like the reference's nn.Linear layers do.
        self.attn_dtype = attn_dtype
        if compute_dtype is not None and compute_dtype not in _lib.DTYPES:
            raise ValueError(f"Unsupported compute_dtype: {compute_dtype}")
        self.compute_dtype = compute_dtype
        # residual_dtype (an extension; 16-bit compute modes only): storage of the residual stream between the layer kernels.
        # "float32" = every residual add, LayerNorm statistic and the head see an fp32 stream (stricter than the reference's
        # autocast inference); "float16" = the stream is held in fp16 like the reference's own "16-mixed" inference holds it
        # (nn.Linear outputs are 16-bit there and layer.py:155-164 adds them), each residual sum formed in fp32 from the fp32
        # accumulators and rounded once -- half the HBM bytes of the two N = 512 GEMMs and the three LayerNorms of a layer.
        # None = RAP_RESIDUAL_DTYPE from the environment, else "auto" = the default chosen by measurement (see DEFAULT_RESIDUAL_DTYPE).
        if residual_dtype is None:
            residual_dtype = os.environ.get("RAP_RESIDUAL_DTYPE") or DEFAULT_RESIDUAL_DTYPE
        if residual_dtype not in ("auto", "float32", "float16"):
            raise ValueError(f"Unsupported residual_dtype: {residual_dtype}")
        self.residual_dtype = residual_dtype
        self.cfg = dict(embed_dim=embed_dim, num_layers=num_layers, num_heads=num_heads, local_feat_dim=local_feat_dim, in_dim=in_dim,
                        qk_norm=self.qk_norm, scale_emb_on=self.scale_emb_on, local_feat_concat_on=self.local_feat_concat_on)
        self._spec = weight_spec(self.cfg)              # the reference's state_dict for THIS configuration (names, shapes, order)
        # the native model always has the full layout: [cond 63 | x_t 63 | scale 21 | feat F'] embedding input and both qk-norm gains;
        # F' = 0 when the features are not concatenated
        self._native_feat = local_feat_dim if self.local_feat_concat_on else 0
        self._native_cfg = dict(embed_dim=embed_dim, num_layers=num_layers, num_heads=num_heads, local_feat_dim=self._native_feat)
        self._sd: dict[str, torch.Tensor] | None = None
        self._handle = ctypes.c_void_p(0)
        self._generation = 0          # bumped whenever the native model is released: captured graphs of an older one must not be replayed
        self._device: torch.device | None = None
        self._desc = _lib.ModelDesc(embed_dim, num_layers, num_heads, self._native_feat)
        lib = _lib.load()
        n = lib.rap_weight_count_latent(ctypes.byref(self._desc), int(in_dim))
        if n < 0:
            raise NotImplementedError(f"unsupported PointCloudDiT configuration {self.cfg}")
        self._n_floats = int(n)

    # ---- weights -------------------------------------------------------------------------------
    def state_dict(self) -> dict[str, torch.Tensor]:
        if self._sd is None:
            raise _lib.RapError("no weights loaded")
        return dict(self._sd)

    def load_state_dict(self, state_dict: dict, strict: bool = True):
        """nn.Module contract (the reference's load_checkpoint_for_module relies on it, utils/checkpoint.py:13-61): returns
        ``(missing_keys, unexpected_keys)``; with ``strict=True`` any missing or unexpected key raises, with ``strict=False``
        missing tensors keep their previous value (the model cannot be used until every tensor has been supplied once)."""
        names = [n for n, _ in self._spec]
        known = set(names)
        missing = [n for n in names if n not in state_dict]
        unexpected = [k for k in state_dict if k not in known]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for PointCloudDiT: missing {missing[:4]}..., "
                               f"unexpected {unexpected[:4]}...")
        sd = dict(self._sd) if self._sd is not None else {}
        for n, shape in self._spec:
            if n not in state_dict:
                continue
            t = state_dict[n]
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {n}: {tuple(t.shape)} vs {shape}")
            sd[n] = t.detach().to(torch.float32)
        self._sd = sd
        self._release()
        return _IncompatibleKeys(missing, unexpected)

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.RapError("rap_amd.PointCloudDiT runs on the GPU only (no CPU fallback)")
        self._ensure_model(device)
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def eval(self):
        return self

    def _release(self):
        if self._handle:
            _lib.load().rap_model_destroy(self._handle)
            self._handle = ctypes.c_void_p(0)
            self._device = None
            self._generation += 1

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _ensure_model(self, device: torch.device):
        if self._handle and self._device == device:
            return
        if self._sd is None:
            raise _lib.RapError("load_state_dict() must be called before the model is used")
        absent = [n for n, _ in self._spec if n not in self._sd]
        if absent:
            raise _lib.RapError(f"weights never loaded for {absent[:4]}... ({len(absent)} tensors): load_state_dict(strict=False) left them out")
        self._release()
        lib = _lib.load()
        with torch.cuda.device(device):
            blob = torch.cat([t.reshape(-1) for t in self._native_tensors()]).to(device=device, dtype=torch.float32)
            assert blob.numel() == self._n_floats
            handle = ctypes.c_void_p(0)
            rc = lib.rap_model_create_latent(ctypes.byref(self._desc), int(self.in_dim), _lib.ptr(blob), blob.numel(),
                                             _lib.current_stream(device), ctypes.byref(handle))
            _lib.check(rc, "rap_model_create")
            if not self.qk_norm:
                _lib.check(lib.rap_model_set_qk_norm(handle, 0), "rap_model_set_qk_norm")
            torch.cuda.current_stream(device).synchronize()   # blob may be freed after this
        self._handle, self._device = handle, device

    def _native_tensors(self):
        """The loaded state_dict in the NATIVE model's layout (rap_amd.synthetic.weight_spec of the full configuration): the embedding
        projection widened with zero columns for inputs this configuration does not concatenate, unit gains where qk_norm is off (the
        native model skips the norm then; the values are never read)."""
        d, H = self.embed_dim, self.num_heads
        for n, shape in weight_spec(self._native_cfg):
            if n in self._sd and tuple(self._sd[n].shape) == tuple(shape) and not (self.in_dim and n == "encoding_manager.emb_proj.weight"):
                yield self._sd[n]
            elif n == "encoding_manager.emb_proj.weight":
                w = self._sd[n]
                F = self._native_feat
                full = torch.zeros((shape[0], 147 + F + self.in_dim), dtype=torch.float32)      # native order: cond | x_t | scale | feat | latent
                full[:, :126] = w[:, :126]                                   # cond PE | x_t PE
                c = 126
                if self.in_dim:                                              # the reference concatenates the latent features HERE (embedding.py:163-166)
                    full[:, 147 + F:] = w[:, c:c + self.in_dim]; c += self.in_dim
                if self.scale_emb_on:
                    full[:, 126:147] = w[:, c:c + 21]; c += 21
                if self.local_feat_concat_on:
                    full[:, 147:147 + self.local_feat_dim] = w[:, c:c + self.local_feat_dim]
                yield full
            elif n.endswith("_norm.gamma") and not self.qk_norm:
                yield torch.ones(shape, dtype=torch.float32)
            else:
                raise _lib.RapError(f"no native layout for {n}")

    def _dtype_code(self) -> int:
        if self.compute_dtype is not None:
            return _lib.DTYPES[self.compute_dtype]
        if torch.is_autocast_enabled("cuda"):
            return {torch.bfloat16: 1, torch.float16: 2}.get(torch.get_autocast_dtype("cuda"), 0)
        return 0

    def _activate(self, device: torch.device):
        """Model resident on `device` with the transformer-block arithmetic type selected; returns the handle."""
        self._ensure_model(device)
        lib = _lib.load()
        code = self._dtype_code()
        resolved = self.residual_dtype if self.residual_dtype != "auto" else ("float16" if code == 1 else "float32")
        rcode = 0 if code == 3 else _lib.DTYPES[resolved]      # split precision keeps the residual stream in fp32
        if lib.rap_model_residual_dtype(self._handle) != rcode:
            _lib.check(lib.rap_model_set_residual_dtype(self._handle, rcode), "rap_model_set_residual_dtype")
        if lib.rap_model_compute_dtype(self._handle) != code:
            with torch.cuda.device(device):
                _lib.check(lib.rap_model_set_compute_dtype(self._handle, code, _lib.current_stream(device)),
                           "rap_model_set_compute_dtype")
        return self._handle

    # ---- forward -------------------------------------------------------------------------------
    def forward(self, x, timesteps, cond_coord, local_features, latent_features, scales, anchor_indices,
                cu_seqlens_batch, cu_seqlens_part, return_transformer_features: bool = False):
        """(TP,3) velocity, or {'velocity','transformer_features'}  (point_cloud_dit.py:141-191)."""
        if (latent_features is None) != (self.in_dim == 0):
            raise ValueError("latent_features must be None (in_dim == 0)" if self.in_dim == 0 else f"latent_features (TP, {self.in_dim}) required")
        _require_cuda(x, "x")
        device = x.device
        self._activate(device)
        lib = _lib.load()
        TP = x.shape[0]
        B = cu_seqlens_batch.shape[0] - 1
        VP = cu_seqlens_part.shape[0] - 1
        x = _f32c(x); cond = _f32c(cond_coord.reshape(TP, 3))
        # local_feat_concat_on=False: the reference ignores the features (embedding.py:175); the native model was built without them
        feats = _f32c(local_features.reshape(TP, -1)) if (self.local_feat_concat_on and local_features is not None) else None
        latent = None if latent_features is None else _f32c(latent_features.to(device).reshape(TP, -1))
        if latent is not None and latent.shape[1] != self.in_dim:
            raise ValueError(f"latent_features must be (TP, {self.in_dim})")
        ts = _f32c(timesteps.to(device)); sc = _f32c(scales.to(device))
        anchor = anchor_indices.to(device=device, dtype=torch.uint8).contiguous()
        cu_b = cu_seqlens_batch.to(device=device, dtype=torch.int32).contiguous()
        cu_p = cu_seqlens_part.to(device=device, dtype=torch.int32).contiguous()
        if (feats is None) != (self._native_feat == 0) or (feats is not None and feats.shape[1] != self.local_feat_dim) or ts.shape[0] != B or sc.shape[0] != B:
            raise ValueError("shape mismatch in PointCloudDiT.forward inputs")
        v = torch.empty((TP, 3), dtype=torch.float32, device=device)
        feat_out = torch.empty((TP, self.embed_dim), dtype=torch.float32, device=device) if return_transformer_features else None
        nbytes = lib.rap_workspace_bytes(self._handle, TP, B, VP, B)
        ws = workspace(device, nbytes)
        with torch.cuda.device(device):
            rc = lib.rap_dit_forward_latent(self._handle, _lib.ptr(x), _lib.ptr(ts), _lib.ptr(cond), _lib.ptr(feats), _lib.ptr(latent), _lib.ptr(sc),
                                            _lib.ptr(anchor), _lib.ptr(cu_b), _lib.ptr(cu_p), B, VP, TP, _lib.ptr(v),
                                            _lib.ptr(feat_out), _lib.ptr(ws), ws.numel(), _lib.current_stream(device))
        _lib.check(rc, "rap_dit_forward")
        if return_transformer_features:
            return {"velocity": v, "transformer_features": feat_out}
        return v

    __call__ = forward
