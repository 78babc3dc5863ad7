"""Rotation of spherical-harmonics coefficients: `rotate_sh(sh_coefficients, rotations)` with the
signature of the reference's src/misc/sh_rotation.py:10-36 - coefficients expressed in a local frame come back expressed in
the frame `rotations` maps to (the encoder uses it with the camera-to-world rotation of a Gaussian's source view).

The reference builds its per-band rotation matrices from e3nn's Wigner-D functions, in e3nn's harmonic basis; e3nn is not
available here, so that exact convention could not be compared.  This implementation works in the basis
the rasterizer EVALUATES (the polynomial real harmonics of csrc/gsr_hip.hip `sh_visit` / oracle `sh_basis`, signs included):
for every band l the (2l+1) x (2l+1) matrix D_l(R) with  sum_k (D_l c)_k Y_k(d) = sum_k c_k Y_k(R^T d)  for all directions d
is obtained exactly (to rounding) by matching both sides on a fixed set of well-spread directions.  Rendering a rotated
scene with rotated coefficients then gives the image of the unrotated one (tests/test_sh_rotation.py)."""
from __future__ import annotations

from functools import lru_cache
from math import isqrt

import torch
from torch import Tensor

_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435)
_C4 = (2.5033429417967046, -1.7701307697799304, 0.9461746957575601, -0.6690465435572892, 0.10578554691520431,
       -0.6690465435572892, 0.47308734787878004, -1.7701307697799304, 0.6258357354491761)


def sh_basis(directions: Tensor, degree: int = 4) -> Tensor:
    """(..., 3) unit directions -> (..., (degree + 1)^2) values of the real harmonics the rasterizer evaluates, in its order."""
    x, y, z = directions.unbind(-1)
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    out = [torch.full_like(x, _C0)]
    if degree > 0:
        out += [-_C1 * y, _C1 * z, -_C1 * x]
    if degree > 1:
        out += [_C2[0] * xy, _C2[1] * yz, _C2[2] * (2 * zz - xx - yy), _C2[3] * xz, _C2[4] * (xx - yy)]
    if degree > 2:
        out += [_C3[0] * y * (3 * xx - yy), _C3[1] * xy * z, _C3[2] * y * (4 * zz - xx - yy), _C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
                _C3[4] * x * (4 * zz - xx - yy), _C3[5] * z * (xx - yy), _C3[6] * x * (xx - 3 * yy)]
    if degree > 3:
        out += [_C4[0] * xy * (xx - yy), _C4[1] * yz * (3 * xx - yy), _C4[2] * xy * (7 * zz - 1), _C4[3] * yz * (7 * zz - 3),
                _C4[4] * (zz * (35 * zz - 30) + 3), _C4[5] * xz * (7 * zz - 3), _C4[6] * (xx - yy) * (7 * zz - 1),
                _C4[7] * xz * (xx - 3 * yy), _C4[8] * (xx * (xx - 3 * yy) - yy * (3 * xx - yy))]
    return torch.stack(out, -1)


@lru_cache(maxsize=None)
def _samples(degree: int):
    """Fibonacci-sphere directions and, per band, the pseudo-inverse of the basis sampled on them (float64, CPU)."""
    p = 96
    k = torch.arange(p, dtype=torch.float64) + 0.5
    z = 1 - 2 * k / p
    phi = k * (torch.pi * (3 - 5 ** 0.5))
    r = (1 - z * z).sqrt()
    dirs = torch.stack((r * phi.cos(), r * phi.sin(), z), -1)
    basis = sh_basis(dirs, degree)
    pinv = [torch.linalg.pinv(basis[:, l * l:(l + 1) * (l + 1)]) for l in range(degree + 1)]
    return dirs, pinv


def band_rotations(rotations: Tensor, degree: int) -> list:
    """(..., 3, 3) -> [ (..., 2l+1, 2l+1) for l = 0 .. degree ]."""
    dirs, pinv = _samples(degree)
    rot = rotations.to(torch.float64)
    local = dirs.to(rot.device) @ rot  # rows: (R^T d)^T for every sample direction d
    basis = sh_basis(local, degree)  # (..., P, n)
    return [pinv[l].to(rot.device) @ basis[..., l * l:(l + 1) * (l + 1)] for l in range(degree + 1)]


def rotate_sh(sh_coefficients: Tensor, rotations: Tensor) -> Tensor:
    """sh_coefficients (*#batch, n), n = (degree + 1)^2 <= 25; rotations (*#batch, 3, 3) -> (*batch, n)."""
    n = sh_coefficients.shape[-1]
    degree = isqrt(n) - 1
    if (degree + 1) ** 2 != n or degree > 4:
        raise ValueError(f"{n} coefficients: expected (degree + 1)^2 with degree <= 4")
    if not torch.allclose(torch.det(rotations), rotations.new_tensor(1)):  # (the reference falls back to no rotation, :21-22)
        return sh_coefficients.broadcast_to((*torch.broadcast_shapes(sh_coefficients.shape[:-1], rotations.shape[:-2]), n))
    mats = band_rotations(rotations, degree)
    out = [(mats[l].to(sh_coefficients.dtype) @ sh_coefficients[..., l * l:(l + 1) * (l + 1), None]).squeeze(-1) for l in range(degree + 1)]
    return torch.cat(out, dim=-1)
