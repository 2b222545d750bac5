"""rap_amd -- MI355X-native (gfx950) rectified-flow registration sampler.

One hot path of PRBonn/RAP, rebuilt as hand-written HIP kernels behind the reference's own Python API:
``rectified_point_flow/{sampler.py, flow_model/, procrustes.py}`` behind
``RectifiedPointFlow.sample_rectified_flow``.  See DESIGN.md / INTEGRATION.md.
"""
from .data import transform_and_collate
from .flow_model import PointCloudDiT
from .modeling import RectifiedPointFlow
from .procrustes import fit_transformations, rigidify_prediction_with_procrustes, solve_procrustes
from .sampler import euler_step, flow_sampler, get_sampler
from .selection import (average_trajectory_rigidity_rmse, compute_overlap_ratio, compute_rigidity_rmse,
                        select_generations_by_overlap, select_generations_by_rigidity)

__all__ = ["PointCloudDiT", "RectifiedPointFlow", "fit_transformations", "rigidify_prediction_with_procrustes",
           "solve_procrustes", "euler_step", "flow_sampler", "get_sampler", "compute_rigidity_rmse",
           "average_trajectory_rigidity_rmse", "select_generations_by_rigidity", "compute_overlap_ratio",
           "select_generations_by_overlap", "transform_and_collate"]
