"""Host-side geometry helpers on the rasterizer hot path.

Counterparts of the reference's pure-torch helpers (same names, argument meaning and results):
  * get_fov                      -- src/geometry/projection.py:233-247
  * homogenize_points            -- src/geometry/projection.py:9-13
  * depth_to_relative_disparity  -- src/model/encoder/costvolume/conversions.py:17-27
  * get_projection_matrix        -- src/model/decoder/cuda_splatting.py:17-44
They are plumbing (tiny batched torch ops on the device the inputs live on); the raster work is in
the HIP library.
"""
from __future__ import annotations

import torch
from torch import Tensor


def homogenize_points(points: Tensor) -> Tensor:
    """(..., d) -> (..., d+1) with a trailing 1 (reference projection.py:9-13)."""
    return torch.cat([points, torch.ones_like(points[..., :1])], dim=-1)


def get_fov(intrinsics: Tensor) -> Tensor:
    """(b,3,3) normalised intrinsics -> (b,2) [fov_x, fov_y] in radians.

    Angle between the un-projected edge-midpoint rays, exactly as the reference does
    (projection.py:233-247); the principal-point offset is therefore not modelled.
    """
    intrinsics_inv = intrinsics.inverse()

    def process_vector(vector):
        vector = torch.tensor(vector, dtype=torch.float32, device=intrinsics.device)
        vector = torch.einsum("bij,j->bi", intrinsics_inv, vector)
        return vector / vector.norm(dim=-1, keepdim=True)

    left = process_vector([0, 0.5, 1])
    right = process_vector([1, 0.5, 1])
    top = process_vector([0.5, 0, 1])
    bottom = process_vector([0.5, 1, 1])
    fov_x = (left * right).sum(dim=-1).acos()
    fov_y = (top * bottom).sum(dim=-1).acos()
    return torch.stack((fov_x, fov_y), dim=-1)


def depth_to_relative_disparity(depth: Tensor, near: Tensor, far: Tensor, eps: float = 1e-10) -> Tensor:
    """Depth -> relative disparity, 0 at near and 1 at far (reference conversions.py:17-27)."""
    disp_near = 1 / (near + eps)
    disp_far = 1 / (far + eps)
    disp = 1 / (depth + eps)
    return 1 - (disp - disp_far) / (disp_near - disp_far + eps)


def get_projection_matrix(near: Tensor, far: Tensor, fov_x: Tensor, fov_y: Tensor) -> Tensor:
    """Symmetric frustum, x/y to (-1,1), z to (0,1), +z forward (reference cuda_splatting.py:17-44)."""
    tan_fov_x = (0.5 * fov_x).tan()
    tan_fov_y = (0.5 * fov_y).tan()

    top = tan_fov_y * near
    bottom = -top
    right = tan_fov_x * near
    left = -right

    (b,) = near.shape
    result = torch.zeros((b, 4, 4), dtype=torch.float32, device=near.device)
    result[:, 0, 0] = 2 * near / (right - left)
    result[:, 1, 1] = 2 * near / (top - bottom)
    result[:, 0, 2] = (right + left) / (right - left)
    result[:, 1, 2] = (top + bottom) / (top - bottom)
    result[:, 3, 2] = 1
    result[:, 2, 2] = far / (far - near)
    result[:, 2, 3] = -(far * near) / (far - near)
    return result
