"""Oracle side of the scale / rotation input form: the covariance PF3plat's encoder builds between its raw outputs and the
decoder, restated with differentiable CPU torch ops (float32 or float64).  TEST INFRASTRUCTURE - only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product builds the covariance inside its
kernels (gsr_forward_scale_rot) and is compared with these functions by the tests.

Pinned to the reference by tests/golden/adapter_fixtures.npz (tests/test_adapter.py): covariances, scales, rotations and
means recorded from the reference's own `GaussianAdapter.forward` / `build_covariance` for seeded inputs.

Follows (file:line under /root/reference):
  * src/model/encoder/common/gaussians.py:8-30    quaternion (x, y, z, w) -> rotation, normalised through 2 / (|q|^2 + eps)
  * src/model/encoder/common/gaussians.py:33-44   Sigma_local = R S S^T R^T
  * src/model/encoder/common/gaussian_adapter.py:79-83   Sigma = C Sigma_local C^T, C = camera-to-world rotation (detached)
"""
from __future__ import annotations

import torch
from torch import Tensor


def rotation_from_quaternion_xyzw(q: Tensor, eps: float = 1e-8) -> Tensor:
    """(..., 4) quaternion x, y, z, w (any length) -> (..., 3, 3)."""
    x, y, z, w = q.unbind(-1)
    t = 2 / ((q * q).sum(-1) + eps)
    rows = [1 - t * (y * y + z * z), t * (x * y - z * w), t * (x * z + y * w),
            t * (x * y + z * w), 1 - t * (x * x + z * z), t * (y * z - x * w),
            t * (x * z - y * w), t * (y * z + x * w), 1 - t * (x * x + y * y)]
    return torch.stack(rows, -1).reshape(*q.shape[:-1], 3, 3)


def covariance_from_scale_rotation(scale_rot: Tensor, frames: Tensor | None = None) -> Tensor:
    """(S, N, 7) records (scale x y z, quaternion x y z w) [+ (S, F, 3, 3) world rotations of the F equal groups] -> (S, N, 3, 3)."""
    s, n, _ = scale_rot.shape
    r = rotation_from_quaternion_xyzw(scale_rot[..., 3:7])
    d = torch.diag_embed(scale_rot[..., 0:3])
    cov = r @ d @ d.transpose(-1, -2) @ r.transpose(-1, -2)
    if frames is not None:
        f = frames.shape[1]
        c = frames.to(cov.dtype).repeat_interleave(n // f, dim=1)
        cov = c @ cov @ c.transpose(-1, -2)
    return cov


def cov6_from_scale_rotation(scale_rot: Tensor, frames: Tensor | None = None) -> Tensor:
    cov = covariance_from_scale_rotation(scale_rot, frames)
    return torch.stack((cov[..., 0, 0], cov[..., 0, 1], cov[..., 0, 2], cov[..., 1, 1], cov[..., 1, 2], cov[..., 2, 2]), -1)
