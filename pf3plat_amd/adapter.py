"""Encoder-side Gaussian adapter: from per-pixel network outputs to the Gaussians the decoder renders - the step right in
front of the raster path.  Counterpart of the reference's `GaussianAdapter`
(src/model/encoder/common/gaussian_adapter.py:30-125; same constructor cfg, same `forward` arguments, `get_scale_multiplier`,
`d_sh`, `d_in`) with one difference in what comes out: the (N, 3, 3) world-space covariances are NOT materialised.  The
adapter returns scales, unit quaternions and the camera-to-world rotation of every source view; the raster library builds
Sigma = (C R) diag(s^2) (C R)^T in registers as it loads a Gaussian (`gsr_forward_scale_rot`) and its backward returns
dL/dscale and dL/dquaternion directly.  That removes the 36 bytes per Gaussian the covariance costs in each direction (written
by the adapter, read by the forward, its gradient written by the backward and read by autograd) and six small torch
kernels.  `AdaptedGaussians.covariances` still yields the matrices (as differentiable torch ops) for callers that want them.

The harmonics are rotated into world space by `pf3plat_amd.sh_rotation.rotate_sh` (the reference uses its e3nn-based
src/misc/sh_rotation.py at gaussian_adapter.py:90); another callable of that signature, or None for no rotation, can be
passed to the constructor.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .sh_rotation import rotate_sh as _rotate_sh
from .types import Gaussians


@dataclass
class GaussianAdapterCfg:
    gaussian_scale_min: float
    gaussian_scale_max: float
    sh_degree: int


@dataclass
class AdaptedGaussians:
    """Per-pixel Gaussians of a batch of scenes, grouped by source view.  Every per-Gaussian tensor is (scene, view, *rest, ...),
    `rest` being whatever batch dimensions the encoder used behind the view axis (the reference calls the adapter with
    (ray, surface, sample), encoder_costvolume.py:529-540; a plain (ray,) works as well) - the shapes the reference's
    `Gaussians` has at that point, so `g.means[:, (0, -1)]`-style indexing reads the same."""

    means: Tensor  # (b, v, *rest, 3) world space
    scales: Tensor  # (b, v, *rest, 3)
    rotations: Tensor  # (b, v, *rest, 4) unit quaternions x, y, z, w, in the source camera's frame
    harmonics: Tensor  # (b, v, *rest, 3, d_sh)
    opacities: Tensor  # (b, v, *rest)
    frames: Tensor  # (b, v, 3, 3) camera-to-world rotation of every source view (no gradient)

    @property
    def covariances(self) -> Tensor:
        """(b, v, *rest, 3, 3) world-space covariances, materialised with torch ops (the raster path does not need them)."""
        x, y, z, w = self.rotations.unbind(-1)
        t = 2 / ((self.rotations * self.rotations).sum(-1) + 1e-8)
        rot = torch.stack((1 - t * (y * y + z * z), t * (x * y - z * w), t * (x * z + y * w),
                           t * (x * y + z * w), 1 - t * (x * x + z * z), t * (y * z - x * w),
                           t * (x * z - y * w), t * (y * z + x * w), 1 - t * (x * x + y * y)), -1).unflatten(-1, (3, 3))
        rest = self.opacities.dim() - 2
        m = self.frames[(slice(None), slice(None)) + (None,) * rest] @ rot
        return (m * (self.scales * self.scales)[..., None, :]) @ m.transpose(-1, -2)

    def for_decoder(self, views: Optional[Sequence[int]] = None) -> Gaussians:
        """One set of Gaussians per scene in the decoder's layout: views flattened into the Gaussian axis (view-major, so the
        Gaussians of a source view are consecutive - the grouping the kernels' `frames` argument expects).  `views`: keep only
        these source views (the reference keeps the first and the last, `[:, (0, -1)]`, encoder_costvolume.py:556-573)."""
        pick = (lambda t: t) if views is None else (lambda t: t[:, list(views)])
        b = self.opacities.shape[0]
        flat = lambda t, k: pick(t).reshape(b, -1, *t.shape[t.dim() - k:])
        return Gaussians(means=flat(self.means, 1), covariances=None, harmonics=flat(self.harmonics, 2),
                         opacities=flat(self.opacities, 0), scales=flat(self.scales, 1), rotations=flat(self.rotations, 1),
                         frames=pick(self.frames))


class GaussianAdapter(nn.Module):
    def __init__(self, cfg: GaussianAdapterCfg, rotate_sh: Optional[Callable[[Tensor, Tensor], Tensor]] = _rotate_sh):
        super().__init__()
        self.cfg = cfg
        self.rotate_sh = rotate_sh
        # band l of the harmonics starts small (0.1 x 0.25^l): the DC term dominates at initialisation
        band = torch.arange(self.d_sh, dtype=torch.float32).sqrt().floor()
        self.register_buffer("sh_mask", torch.where(band == 0, torch.ones(()), 0.1 * 0.25 ** band), persistent=False)

    @property
    def d_sh(self) -> int:
        return (self.cfg.sh_degree + 1) ** 2

    @property
    def d_in(self) -> int:
        return 7 + 3 * self.d_sh

    def get_scale_multiplier(self, intrinsics: Tensor, pixel_size: Tensor, multiplier: float = 0.1) -> Tensor:
        """How large one pixel is at unit depth, summed over x and y: (K[:2, :2]^-1 pixel_size) . (1, 1), times `multiplier`."""
        return multiplier * torch.linalg.solve(intrinsics[..., :2, :2], pixel_size.expand(*intrinsics.shape[:-2], 2)).sum(-1)

    def forward(self, extrinsics: Tensor, intrinsics: Tensor, coordinates: Tensor, depths: Tensor, opacities: Tensor,
                raw_gaussians: Tensor, image_shape: tuple, eps: float = 1e-8) -> AdaptedGaussians:
        """The reference's `*#batch` call shapes (gaussian_adapter.py:48-58): every argument broadcasts against
        `opacities.shape` = (b, v, *rest).  extrinsics (b, v, 1.., 4, 4) / intrinsics (b, v, 1.., 3, 3) of the source views (one
        camera per view: singleton dims behind the view axis), coordinates (b, v, *rest#, 2) in [0, 1]^2, depths (b, v, *rest#),
        raw_gaussians (b, v, *rest#, 7 + 3 d_sh) = scale(3) | quaternion xyzw(4) | harmonics(3 x d_sh).  E.g. the encoder's call
        (encoder_costvolume.py:529-540): extrinsics (b, v, 1, 1, 1, 4, 4), coordinates (b, v, r, srf, 1, 2), depths (b, v, r, 1, 1),
        opacities (b, v, r, srf, spp), raw (b, v, r, srf, 1, c)."""
        h, w = image_shape
        full = tuple(opacities.shape)  # (b, v, *rest)
        if len(full) < 3:
            raise ValueError(f"opacities must be (scene, view, ray, ...), got {full}")
        if any(d != 1 for d in extrinsics.shape[2:-2]) or tuple(extrinsics.shape[:2]) != full[:2]:
            raise ValueError("one camera per (scene, view): extrinsics must be (b, v, 1.., 4, 4); a rotation per Gaussian cannot be "
                             "expressed as the per-view frames the raster kernels take")
        raw_scale, raw_quat, raw_sh = raw_gaussians.split((3, 4, 3 * self.d_sh), dim=-1)
        lo, hi = self.cfg.gaussian_scale_min, self.cfg.gaussian_scale_max
        pixel = torch.tensor((1.0 / w, 1.0 / h), dtype=torch.float32, device=extrinsics.device)
        footprint = depths * self.get_scale_multiplier(intrinsics, pixel)  # world size of 0.1 pixel at the Gaussian's depth
        scales = (lo + (hi - lo) * raw_scale.sigmoid()) * footprint[..., None]
        rotations = raw_quat / (raw_quat.norm(dim=-1, keepdim=True) + eps)
        harmonics = raw_sh.unflatten(-1, (3, self.d_sh)).broadcast_to((*full, 3, self.d_sh)) * self.sh_mask
        c2w = extrinsics[..., :3, :3].detach()
        # mean = camera centre + depth x (unit ray through the pixel, rotated into world space)
        ray = torch.linalg.solve(intrinsics, F.pad(coordinates, (0, 1), value=1.0).unsqueeze(-1)).squeeze(-1)
        ray = F.normalize(ray, dim=-1)
        direction = (extrinsics[..., :3, :3] @ ray.unsqueeze(-1)).squeeze(-1)
        means = extrinsics[..., :3, 3] + direction * depths[..., None]
        if self.rotate_sh is not None:
            harmonics = self.rotate_sh(harmonics, c2w[..., None, :, :])
        return AdaptedGaussians(means=means.broadcast_to((*full, 3)), scales=scales.broadcast_to((*full, 3)),
                                rotations=rotations.broadcast_to((*full, 4)), harmonics=harmonics, opacities=opacities,
                                frames=c2w.reshape(*full[:2], 3, 3))
