"""Deterministic PF3plat-shaped synthetic scenes (SURVEY.md §8d "Synthetic scene generator").

There is no dataset and no encoder in this build; the bench and the parity tests need Gaussians
whose screen-space footprint distribution looks like what PF3plat's `GaussianAdapter` emits
(reference src/model/encoder/common/gaussian_adapter.py:48-111): pixel-aligned Gaussians
un-projected from two context cameras, scale proportional to depth x pixel size, random unit
quaternions, sigmoid opacities and SH coefficients masked towards the DC band.

All randomness comes from one CPU `torch.Generator`, so a (seed, N, H, W) tuple names one scene
on every machine; tensors are moved to `device` afterwards.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
from torch import Tensor

from .types import Gaussians


@dataclass
class Scene:
    gaussians: Gaussians  # batch = 1
    extrinsics: Tensor  # (1, v, 4, 4) camera-to-world, OpenCV convention
    intrinsics: Tensor  # (1, v, 3, 3) normalised
    near: Tensor  # (1, v)
    far: Tensor  # (1, v)
    image_shape: tuple
    background: Tensor  # (3,)

    def to(self, device) -> "Scene":
        g = self.gaussians
        return Scene(
            Gaussians(g.means.to(device), g.covariances.to(device), g.harmonics.to(device), g.opacities.to(device)),
            self.extrinsics.to(device), self.intrinsics.to(device), self.near.to(device), self.far.to(device),
            self.image_shape, self.background.to(device),
        )


def _quat_xyzw_to_matrix(q: Tensor) -> Tensor:
    i, j, k, r = q.unbind(-1)
    two_s = 2 / (q * q).sum(-1)
    o = torch.stack(
        (1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
         two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
         two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(*q.shape[:-1], 3, 3)


def _smooth_depth_surface(g: torch.Generator, hs: int, ws: int) -> Tensor:
    """(hs, ws) depth map of one context view: a tilted ground plane meeting a back wall, two fronto-parallel slabs standing in
    front of it (depth discontinuities, as objects make them) and a low-frequency ripple - what a depth head emits for an indoor
    frame, not noise.  Depths in [1.2, 20] in units of `near`."""
    ys = (torch.arange(hs, dtype=torch.float32) + 0.5) / hs
    xs = (torch.arange(ws, dtype=torch.float32) + 0.5) / ws
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
    r = torch.rand(12, generator=g, dtype=torch.float32)
    wall = 8.0 + 8.0 * r[0] + (xx - 0.5) * (4.0 * r[1] - 2.0)                # back wall, slightly turned
    horizon = 0.45 + 0.1 * r[2]
    ground = (1.4 + 0.8 * r[3]) / (yy - horizon).clamp_min(1e-3)             # ground plane below the horizon
    d = torch.minimum(wall, torch.where(yy > horizon, ground, torch.full_like(ground, 1e9)))
    for j in range(2):                                                        # two slabs in front
        cx, cy = 0.2 + 0.6 * r[4 + 2 * j], 0.35 + 0.4 * r[5 + 2 * j]
        half_w, half_h = 0.08 + 0.12 * r[8 + j], 0.12 + 0.15 * r[10 + j]
        inside = ((xx - cx).abs() < half_w) & ((yy - cy).abs() < half_h)
        d = torch.where(inside, torch.minimum(d, torch.full_like(d, 2.5 + 3.0 * j + 2.0 * float(r[4 + j]))), d)
    ph = 6.283185 * torch.rand(4, generator=g, dtype=torch.float32)
    ripple = 0.03 * (torch.sin(6.283185 * 1.5 * xx + ph[0]) * torch.sin(6.283185 * 1.0 * yy + ph[1])
                     + 0.5 * torch.sin(6.283185 * 3.0 * xx + ph[2]) * torch.sin(6.283185 * 2.5 * yy + ph[3]))
    return (d * (1.0 + ripple)).clamp(1.2, 20.0)


def make_scene(seed: int, num_gaussians: int, image_shape=(256, 256), d_sh: int = 25, num_views: int = 1,
               view_offsets=None, near: float = 1.0, far: float = 100.0, device="cpu", structure: str = "random",
               source_shape=None) -> Scene:
    """Seeded scene: Gaussians + `num_views` render cameras.

    Render camera 0 is c2w = I; further views are shifted along x by `view_offsets` (default: evenly
    spaced in [-0.25, 0.25], i.e. between the two source cameras at x = -0.5 / +0.5).

    structure = "random" (SURVEY 8d): pixel, depth (log-uniform 1 .. 20), scale and opacity drawn independently per Gaussian.
    structure = "pixel_aligned": what PF3plat's encoder hands its decoder (reference src/model/encoder/encoder_costvolume.py:509-573,
    src/model/encoder/common/gaussian_adapter.py:63-111): ONE Gaussian per pixel of each of the two context images, in raster
    order - index = view * hs * ws + row * ws + col - its centre on the pixel's ray (sub-pixel offset sigmoid(N(0,1)) - 0.5 pixels,
    :515-517) at the depth of a smooth per-view surface (+ 0.5 % noise), scale = (0.5 + 14.5 sigmoid) x depth x the adapter's pixel
    multiplier (:100-111), opacity = the density head's pdf (map_pdf_to_opacity with exponent 1, :174-187) skewed towards 1.
    `source_shape` = (hs, ws) of the context images, 2 hs ws = num_gaussians (default: square).
    """
    g = torch.Generator().manual_seed(int(seed))
    h, w = image_shape
    n = int(num_gaussians)
    f = 0.86
    k = torch.tensor([[f, 0, 0.5], [0, f, 0.5], [0, 0, 1]], dtype=torch.float32)

    def rand(*s):
        return torch.rand(*s, generator=g, dtype=torch.float32)

    def randn(*s):
        return torch.randn(*s, generator=g, dtype=torch.float32)

    if structure not in ("random", "pixel_aligned"):
        raise ValueError(f"unknown scene structure {structure!r}")
    # two "context" cameras, half the Gaussians each
    src_x = torch.where(torch.arange(n) < n // 2, -0.5, 0.5).to(torch.float32)
    if structure == "pixel_aligned":
        if source_shape is None:
            side = math.isqrt(n // 2)
            source_shape = (side, side)
        hs, ws = source_shape
        if 2 * hs * ws != n:
            raise ValueError(f"pixel_aligned scene: num_gaussians {n} is not 2 x {hs} x {ws} (one Gaussian per context pixel)")
        rows_, cols_ = torch.meshgrid(torch.arange(hs, dtype=torch.float32), torch.arange(ws, dtype=torch.float32), indexing="ij")
        off = torch.sigmoid(randn(n, 2)) - 0.5
        u = torch.stack(((cols_.reshape(-1).repeat(2) + 0.5 + off[:, 0]) / ws, (rows_.reshape(-1).repeat(2) + 0.5 + off[:, 1]) / hs), -1)
        depth = torch.cat([_smooth_depth_surface(g, hs, ws).reshape(-1) for _ in range(2)]) * (1.0 + 0.005 * randn(n))
        m = 0.1 * (1.0 / (f * ws) + 1.0 / (f * hs))
    else:
        u = rand(n, 2)
    ray = torch.stack(((u[:, 0] - 0.5) / f, (u[:, 1] - 0.5) / f, torch.ones(n)), -1)
    ray = ray / ray.norm(dim=-1, keepdim=True)
    if structure == "random":
        depth = torch.exp(rand(n) * math.log(20.0))
        m = 0.1 * (1.0 / (f * w) + 1.0 / (f * h))
    means = ray * depth[:, None]
    means[:, 0] += src_x
    scales = (0.5 + 14.5 * torch.sigmoid(randn(n, 3))) * depth[:, None] * m
    quat = randn(n, 4)
    quat = quat / quat.norm(dim=-1, keepdim=True)
    rot = _quat_xyzw_to_matrix(quat)
    cov = rot @ torch.diag_embed(scales * scales) @ rot.transpose(-1, -2)
    cov = 0.5 * (cov + cov.transpose(-1, -2))
    opac = torch.sigmoid(randn(n)) if structure == "random" else torch.sigmoid(2.5 + 1.5 * randn(n))
    deg = math.isqrt(d_sh) - 1
    mask = torch.ones(d_sh)
    for l in range(1, deg + 1):
        mask[l * l:(l + 1) * (l + 1)] = 0.1 * 0.25 ** l
    sh = randn(n, 3, d_sh) * mask

    if view_offsets is None:
        view_offsets = [0.0] if num_views == 1 else [(-0.25 + 0.5 * i / (num_views - 1)) for i in range(num_views)]
    assert len(view_offsets) == num_views
    c2w = torch.eye(4).repeat(num_views, 1, 1)
    for i, ox in enumerate(view_offsets):
        c2w[i, 0, 3] = float(ox)
    scene = Scene(
        Gaussians(means[None], cov[None], sh[None], opac[None]),
        c2w[None], k.repeat(1, num_views, 1, 1),
        torch.full((1, num_views), float(near)), torch.full((1, num_views), float(far)),
        (h, w), torch.zeros(3),
    )
    return scene.to(device)


def scene_operator_inputs(scene: Scene, use_sh: bool = True):
    """Scene (batch 1) -> the raw operator tensors of `rasterize_views`:
    means (1,N,3), cov6 (1,N,6), opacities (1,N), colors (1,N,M,3) [or (1,N,3) DC-only when use_sh is False]."""
    g = scene.gaussians
    cov = g.covariances
    cov6 = torch.stack((cov[..., 0, 0], cov[..., 0, 1], cov[..., 0, 2], cov[..., 1, 1], cov[..., 1, 2], cov[..., 2, 2]), -1)
    colors = g.harmonics.permute(0, 1, 3, 2).contiguous()
    if not use_sh:
        colors = colors[:, :, 0, :].contiguous()
    return g.means.contiguous(), cov6.contiguous(), g.opacities.contiguous(), colors


def scene_viewbuf(scene: Scene, scale_invariant: bool = True) -> Tensor:
    """Camera records (V, 48) of a Scene on the raster backend's device: one `setup_views` launch, as `render_cuda` does."""
    from .rasterizer import get_backend

    s, v = scene.extrinsics.shape[:2]
    backend = get_backend()
    dev = scene.extrinsics.device if scene.extrinsics.is_cuda else getattr(backend, "default_device", scene.extrinsics.device)
    mv = lambda t: t.to(dev)
    return backend.setup_views(mv(scene.extrinsics.reshape(s * v, 4, 4)), mv(scene.intrinsics.reshape(s * v, 3, 3)),
                               mv(scene.near.reshape(s * v)), mv(scene.far.reshape(s * v)), mv(scene.background.reshape(3)),
                               scale_invariant)
