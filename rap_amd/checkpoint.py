"""Host-side mirror of the reference's checkpoint loader for the modules rap_amd replaces
(``rectified_point_flow/utils/checkpoint.py:13-61``; MiniSpinNet's ``Desc.`` filter, ``extract_sample_features.py:127-136``).

Weights enter the C ABI as a flat fp32 blob in ``state_dict`` order; these helpers do the key editing the reference does before
``load_state_dict`` so that a Lightning checkpoint (keys ``flow_model.*``) or a ``mini_spinnet_t.pth`` (keys ``Desc.*``) can be
handed over unchanged.
"""
from __future__ import annotations

import torch


def load_checkpoint_for_module(module, checkpoint_path: str, prefix_to_remove: str | None = None,
                               keys_to_substitute: dict | None = None, prefix_to_add: str | None = None, strict: bool = False):
    """Same signature and key-editing order as the reference: read ``ckpt["state_dict"]``, keep and strip ``prefix_to_remove``,
    substitute prefixes, add a prefix, then ``module.load_state_dict(..., strict=strict)``.  ``module`` is a
    ``rap_amd.PointCloudDiT`` / ``rap_amd.spinnet.MiniSpinNet`` (or anything with ``load_state_dict``)."""
    state_dict = torch.load(checkpoint_path, map_location="cpu", weights_only=False)["state_dict"]
    if prefix_to_remove is not None:
        state_dict = {k.replace(prefix_to_remove, ""): v for k, v in state_dict.items() if k.startswith(prefix_to_remove)}
    if keys_to_substitute is not None:
        for old_prefix, new_prefix in keys_to_substitute.items():
            state_dict = {k.replace(old_prefix, new_prefix): v for k, v in state_dict.items()}
    if prefix_to_add is not None:
        state_dict = {f"{prefix_to_add}{k}": v for k, v in state_dict.items()}
    return module.load_state_dict(state_dict, strict=strict)


def load_spinnet_checkpoint(module, checkpoint_path: str):
    """``mini_spinnet_t.pth`` as extract_sample_features.py:120-136 reads it: a flat dict whose ``Desc.*`` entries are the
    MiniSpinNet weights (prefix stripped), loaded non-strictly."""
    state_dict = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    filtered = {k[5:]: v for k, v in state_dict.items() if k.startswith("Desc.")}
    return module.load_state_dict(filtered, strict=False)
