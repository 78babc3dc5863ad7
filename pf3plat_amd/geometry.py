"""Small closed-form camera helpers with the names the reference exposes
(`get_fov` src/geometry/projection.py:233-247, `homogenize_points` :9-13, `depth_to_relative_disparity`
src/model/encoder/costvolume/conversions.py:17-27, `get_projection_matrix` src/model/decoder/cuda_splatting.py:17-44).

They are NOT on the render path: the raster library builds its camera records on the device (`gsr_setup_views`,
`gsr_setup_views_orthographic`).  They exist for callers that want the same quantities as tensors, and are checked against
values recorded from the reference (tests/test_wrapper_fixtures.py, case G).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor

_EDGE_MIDPOINTS = ((0.0, 0.5), (1.0, 0.5), (0.5, 0.0), (0.5, 1.0))  # left, right, top, bottom of the unit image


def homogenize_points(points: Tensor) -> Tensor:
    """(..., d) -> (..., d + 1): a 1 appended to every point."""
    return F.pad(points, (0, 1), value=1.0)


def get_fov(intrinsics: Tensor) -> Tensor:
    """(b, 3, 3) normalised intrinsics -> (b, 2) [fov_x, fov_y] in radians: the angle subtended by the mid-points of opposite
    image edges, seen from the optical centre (so an off-centre principal point barely registers - as in the reference)."""
    pts = homogenize_points(torch.tensor(_EDGE_MIDPOINTS, dtype=intrinsics.dtype, device=intrinsics.device))  # (4, 3)
    rays = torch.linalg.solve(intrinsics, pts.T.expand(intrinsics.shape[0], 3, 4))  # K^-1 p for the four points: (b, 3, 4)
    rays = F.normalize(rays, dim=1)
    cosines = (rays[:, :, 0::2] * rays[:, :, 1::2]).sum(dim=1)  # (left . right, top . bottom)
    return cosines.acos()


def depth_to_relative_disparity(depth: Tensor, near: Tensor, far: Tensor, eps: float = 1e-10) -> Tensor:
    """Depth -> disparity rescaled so that `near` maps to 0 and `far` to 1."""
    inv = lambda x: (x + eps).reciprocal()
    return 1 - (inv(depth) - inv(far)) / (inv(near) - inv(far) + eps)


def get_projection_matrix(near: Tensor, far: Tensor, fov_x: Tensor, fov_y: Tensor) -> Tensor:
    """(b,) x 4 -> (b, 4, 4): symmetric frustum, x / y to (-1, 1), z to (0, 1), +z forward.  Closed form:
    diag(1 / tan(fov_x / 2), 1 / tan(fov_y / 2), f / (f - n), 0) with P[2, 3] = -f n / (f - n) and P[3, 2] = 1."""
    depth_range = far - near
    zero, one = torch.zeros_like(near), torch.ones_like(near)
    rows = (
        ((0.5 * fov_x).tan().reciprocal(), zero, zero, zero),
        (zero, (0.5 * fov_y).tan().reciprocal(), zero, zero),
        (zero, zero, far / depth_range, -(far * near) / depth_range),
        (zero, zero, one, zero),
    )
    return torch.stack([torch.stack(r, dim=-1) for r in rows], dim=-2).to(torch.float32)
