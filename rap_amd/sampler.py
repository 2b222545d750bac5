"""Host-side mirror of ``rectified_point_flow/sampler.py``.

``get_sampler('euler')`` returns a sampler with the reference signature (sampler.py:11-24,154-171)
that accepts ANY ``flow_model_fn(x, t) -> v`` callable; the tensor updates run in librapflow
(``rap_euler_step``, ``rap_rigidify_blend``), not in PyTorch ops.  When the velocity network is
``rap_amd.PointCloudDiT`` use ``RectifiedPointFlow.sample_rectified_flow`` instead: it runs the
whole loop inside one C call (``rap_sample``).
"""
from __future__ import annotations

from functools import partial
from typing import Callable

import torch

from . import _lib
from .flow_model import _f32c, _require_cuda
from .procrustes import rigidify_blend


def euler_step(x_t: torch.Tensor, t: float, dt: float, flow_model_fn: Callable, anchor_indices, x_0):
    """sampler.py:79-92 -> (x_t_next, x_0_hat).  anchor_indices / x_0 are accepted and unused (anchor-free)."""
    v = _f32c(flow_model_fn(x_t, t))
    x_t = _f32c(x_t)
    x0_hat = torch.empty_like(x_t)
    x_next = torch.empty_like(x_t)
    lib = _lib.load()
    with torch.cuda.device(x_t.device):
        rc = lib.rap_euler_step(_lib.ptr(x_t), _lib.ptr(v), float(t), float(dt), _lib.ptr(x0_hat), _lib.ptr(x_next),
                                _lib.ptr(None), x_t.numel(), _lib.current_stream(x_t.device))
    _lib.check(rc, "rap_euler_step")
    return x_next, x0_hat


def flow_sampler(step_fn: Callable, flow_model_fn: Callable, x_1: torch.Tensor, x_0: torch.Tensor,
                 anchor_indices: torch.Tensor, num_steps: int = 20, points_per_part: torch.Tensor | None = None,
                 cu_seqlens_batch: torch.Tensor | None = None, condition: torch.Tensor | None = None,
                 return_trajectory: bool = False, rigidity_forcing: bool = False,
                 return_end_point_trajectory: bool = True):
    """sampler.py:11-74.  t runs 1 -> dt in Python doubles; the raw x0_hat is what the trajectory stores."""
    _require_cuda(x_1, "x_1")
    dt = 1.0 / num_steps
    x_t = _f32c(x_1).clone()
    x_1 = _f32c(x_1)
    if return_trajectory:
        trajectory = torch.empty((num_steps, *x_1.shape), device=x_1.device)
        trajectory_x_t = torch.empty((num_steps, *x_1.shape), device=x_1.device)
    for step in range(num_steps):
        t = 1 - step * dt
        x_t, x_0_hat = step_fn(x_t, t, dt, flow_model_fn, anchor_indices, x_0)
        if rigidity_forcing:
            if cu_seqlens_batch is None and x_0_hat.ndim == 2:
                raise ValueError("cu_seqlens_batch is required when pointclouds has shape (TP, 3)")
            x_t = rigidify_blend(x_0_hat, condition, points_per_part, x_1, 1 - t + dt, t - dt)
        if return_trajectory:
            trajectory[step].copy_(x_0_hat)
            trajectory_x_t[step].copy_(x_t)
    if return_trajectory:
        return {"end_point_trajectory": trajectory, "trajectory": trajectory_x_t}
    return x_t.detach()


def get_sampler(sampler_name: str):
    """sampler.py:154-171: only 'euler' is enabled in the reference (rk2/rk4 are commented out)."""
    step_fns = {"euler": euler_step}
    if sampler_name not in step_fns:
        raise ValueError(f"Unknown sampler: {sampler_name}. Available: {list(step_fns.keys())}")
    return partial(flow_sampler, step_fns[sampler_name])
