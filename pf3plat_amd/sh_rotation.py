"""Rotation of spherical-harmonics coefficients: `rotate_sh(sh_coefficients, rotations)` with the
signature of the reference's src/misc/sh_rotation.py:10-36 (the encoder's adapter calls it with the camera-to-world rotation
of a Gaussian's source view, gaussian_adapter.py:90-92).

Two conventions, selected by `basis`:

* ``"e3nn"`` (default - what the REFERENCE computes).  The reference multiplies band l by e3nn's
  `wigner_D(l, *matrix_to_angles(R))`, the representation matrix in E3NN's real harmonic basis, although the rasterizer then
  evaluates the coefficients in ITS basis (the reference's own ply_export.py:75-77 calls the axes "swizzled").  Weights trained
  with the reference therefore only reproduce the reference's colours if that same matrix is applied.  e3nn is not installed
  here, so the matrix is rebuilt from e3nn's published conventions instead of imported:
    - e3nn's real harmonics Y_e are the standard real harmonics W WITHOUT Condon-Shortley phase (all leading coefficients
      positive: `sh_1 = (x, y, z)`, `sh_2_0 = sqrt(15) x z`, ..., `sh_2_4 = sqrt(15)/2 (z^2 - x^2)`) with the y axis as polar
      axis:  Y_e(x, y, z) = W(z, x, y)  (component order m = -l..l);
    - `D(R) Y_e(d) = Y_e(R d)` is the matrix `wigner_D(l, *matrix_to_angles(R))` (e3nn's equivariance statement).
  The rasterizer's basis B (csrc/gsr_hip.hip `sh_visit`) is W WITH the phase: B_lm = (-1)^m W_lm.  Hence, with the permutation
  P: (x, y, z) -> (z, x, y) and S = diag((-1)^m):   D_e3nn(R) = S D_B(P R P^T) S,   D_B being the exact representation matrix in
  the rasterizer's basis that the "rasterizer" mode builds.  **Unpinned**: no e3nn output could be recorded in this container;
  tests/test_sh_rotation.py checks the consequences that can be checked without it (band 1 equals R itself in (x, y, z) order as
  e3nn documents for its l = 1 irrep, the equivariance identity against e3nn's generated polynomial forms for l <= 3, the group
  law, orthogonality).
* ``"rasterizer"``: rotation in the basis the rasterizer evaluates, i.e. the physically consistent one: rendering a rotated
  scene with coefficients rotated this way gives the image of the unrotated scene (tests/test_sh_rotation.py).  Use it for
  scenes that were NOT trained through the reference's adapter (external .ply files, synthetic data).

For every band l the (2l+1) x (2l+1) matrix D_B(R) with  D_B(R) B(d) = B(R d)  for all directions d is obtained exactly (to
rounding) by matching both sides on a fixed set of well-spread directions."""
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


_P_E3NN = ((0.0, 0.0, 1.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0))  # (x, y, z) -> (z, x, y): e3nn's polar axis is y


def _phase_signs(l: int, device) -> Tensor:
    """(-1)^m for m = -l..l: the Condon-Shortley phase the rasterizer's basis carries and e3nn's does not."""
    m = torch.arange(-l, l + 1, device=device)
    return 1.0 - 2.0 * (m % 2).to(torch.float64)


def e3nn_band_rotations(rotations: Tensor, degree: int) -> list:
    """(..., 3, 3) -> [ (..., 2l+1, 2l+1) ]: what e3nn's `wigner_D(l, *matrix_to_angles(R))` returns (module docstring)."""
    p = torch.tensor(_P_E3NN, dtype=torch.float64, device=rotations.device)
    mats = band_rotations(p @ rotations.to(torch.float64) @ p.T, degree)
    out = []
    for l, m in enumerate(mats):
        sg = _phase_signs(l, rotations.device)
        out.append(sg[:, None] * m * sg[None, :])
    return out


# Harmonics that did NOT come out of the reference's adapter: `ply_export.gaussians_from_ply` tags the tensor it returns (an attribute
# on the tensor object: `_pf3plat_external_harmonics`) and registers its storage here BY WEAK REFERENCE to the tensor - views of it (a
# scene, a slice, a permutation) share the storage, so `rotate_sh` can say when the reference's e3nn-convention default is about to be
# applied to coefficients that live in the rasterizer's basis.  The entry disappears with the tensor: an address the caching allocator
# hands out again later cannot trigger (and use up) the warning, and the table does not grow.  (Copies - `.to()`, `.clone()` - are new
# storages and are not tracked: pass `basis=` explicitly for those.)
_EXTERNAL_STORAGES: dict = {}  # storage address -> weakref to the tagged tensor
_warned_external = False


def mark_external_harmonics(harmonics: Tensor) -> Tensor:
    import weakref

    addr = harmonics.untyped_storage().data_ptr()
    harmonics._pf3plat_external_harmonics = True
    _EXTERNAL_STORAGES[addr] = weakref.ref(harmonics, lambda _r, a=addr: _EXTERNAL_STORAGES.pop(a, None))
    return harmonics


def _is_external(t: Tensor) -> bool:
    if getattr(t, "_pf3plat_external_harmonics", False):
        return True
    ref = _EXTERNAL_STORAGES.get(t.untyped_storage().data_ptr())
    return ref is not None and ref() is not None


def rotate_sh(sh_coefficients: Tensor, rotations: Tensor, basis: str | None = None) -> Tensor:
    """sh_coefficients (*#batch, n), n = (degree + 1)^2 <= 25; rotations (*#batch, 3, 3) -> (*batch, n).
    basis: "e3nn" (the default when None) = the matrices the reference applies (src/misc/sh_rotation.py:24-34), "rasterizer" = the
    rotation in the basis the rasterizer evaluates (see the module docstring).  Called WITHOUT a basis on harmonics that came from
    `gaussians_from_ply` (an external scene: its coefficients were never rotated by the reference's adapter) it warns once - the e3nn
    convention does not keep such a scene's colours still under a rotation, `basis="rasterizer"` does."""
    global _warned_external
    if basis is None:
        basis = "e3nn"
        if not _warned_external and _EXTERNAL_STORAGES and sh_coefficients.numel() > 0 and _is_external(sh_coefficients):
            import warnings

            _warned_external = True
            warnings.warn("rotate_sh: these harmonics come from gaussians_from_ply (an external scene); the default basis='e3nn' "
                          "reproduces the reference's adapter for ITS OWN weights and does not keep an external scene's colours "
                          "under a rotation - pass basis='rasterizer' (or basis='e3nn' to silence this).", UserWarning, stacklevel=2)
    n = sh_coefficients.shape[-1]
    degree = isqrt(n) - 1
    if (degree + 1) ** 2 != n or degree > 4:
        raise ValueError(f"{n} coefficients: expected (degree + 1)^2 with degree <= 4")
    if basis not in ("e3nn", "rasterizer"):
        raise ValueError(f"basis must be 'e3nn' or 'rasterizer', got {basis!r}")
    if not torch.allclose(torch.det(rotations), rotations.new_tensor(1)):  # (the reference falls back to no rotation, :21-22)
        return sh_coefficients.broadcast_to((*torch.broadcast_shapes(sh_coefficients.shape[:-1], rotations.shape[:-2]), n))
    mats = e3nn_band_rotations(rotations, degree) if basis == "e3nn" else band_rotations(rotations, degree)
    out = [(mats[l].to(sh_coefficients.dtype) @ sh_coefficients[..., l * l:(l + 1) * (l + 1), None]).squeeze(-1) for l in range(degree + 1)]
    return torch.cat(out, dim=-1)


def rotate_sh_rasterizer_basis(sh_coefficients: Tensor, rotations: Tensor) -> Tensor:
    return rotate_sh(sh_coefficients, rotations, basis="rasterizer")
