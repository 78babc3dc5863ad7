"""Host wrappers of the rasterizer: counterparts of the reference's
src/model/decoder/cuda_splatting.py (`render_cuda` :47-127, `render_cuda_orthographic` :130-220,
`render_depth_cuda` :226-269) with the same names, argument meaning and results.

What differs is how the device is driven, not what is computed:
  * the reference loops over views in Python, with two `.item()` host syncs and one rasterizer
    invocation per view (:91-126); here all `batch` views go through ONE launch chain
    (`rasterize_views`), camera parameters never leave the device;
  * the scale-invariant pre-scale of means/covariances (:64-71) is not materialised: the factor
    travels in the per-view camera record and is applied on load inside the kernels (same fp32
    multiplies), and its chain rule is applied inside the backward kernel;
  * `render_views` (below) additionally shares one copy of the Gaussians between all views of a
    scene and blends the depth image as a 4th channel of the same pass.
"""
from __future__ import annotations

from math import isqrt
from typing import Optional

import torch
from torch import Tensor

from .geometry import depth_to_relative_disparity, get_fov, get_projection_matrix, homogenize_points
from .rasterizer import get_backend, pack_views, rasterize_views
from .types import DepthRenderingMode

def _cameras(extrinsics, intrinsics, near, far, scale_invariant: bool):
    """Shared camera set-up of render_cuda (reference :64-71, :80-87): returns per-view
    (view_matrix^T, full_projection^T, campos, tan_fov_x, tan_fov_y, scale)."""
    if scale_invariant:
        scale = 1 / near
        extrinsics = extrinsics.clone()
        extrinsics[..., :3, 3] = extrinsics[..., :3, 3] * scale[:, None]
        near = near * scale
        far = far * scale
    else:
        scale = torch.ones_like(near)
    fov_x, fov_y = get_fov(intrinsics).unbind(dim=-1)
    tan_fov_x = (0.5 * fov_x).tan()
    tan_fov_y = (0.5 * fov_y).tan()
    projection_matrix = get_projection_matrix(near, far, fov_x, fov_y).transpose(-1, -2)
    view_matrix = extrinsics.inverse().transpose(-1, -2)
    full_projection = view_matrix @ projection_matrix
    return view_matrix, full_projection, extrinsics[:, :3, 3], tan_fov_x, tan_fov_y, scale


def _viewbuf(extrinsics, intrinsics, near, far, background_color, scale_invariant: bool) -> Tensor:
    """Camera records for `rasterize_views`: one `gsr_setup_views` launch when the backend has it (the HIP library), else the
    batched torch ops above (the arithmetic of the reference wrapper; used by the CPU tests)."""
    with torch.no_grad():
        backend = get_backend()
        if hasattr(backend, "setup_views"):
            return backend.setup_views(extrinsics, intrinsics, near, far, background_color, scale_invariant)
        view_matrix, full_projection, campos, tan_x, tan_y, scale = _cameras(extrinsics, intrinsics, near, far, scale_invariant)
        bg = background_color if background_color.dim() == 2 else background_color.reshape(1, 3).expand(extrinsics.shape[0], 3)
        return pack_views(view_matrix, full_projection, campos, tan_x, tan_y, bg, scale, near=near, far=far)


def render_cuda(
    extrinsics: Tensor,  # (batch, 4, 4) camera-to-world
    intrinsics: Tensor,  # (batch, 3, 3) normalised
    near: Tensor,  # (batch,)
    far: Tensor,  # (batch,)
    image_shape: tuple,
    background_color: Tensor,  # (batch, 3)
    gaussian_means: Tensor,  # (batch, gaussian, 3)
    gaussian_covariances: Tensor,  # (batch, gaussian, 3, 3)
    gaussian_sh_coefficients: Tensor,  # (batch, gaussian, 3, d_sh)
    gaussian_opacities: Tensor,  # (batch, gaussian)
    scale_invariant: bool = True,
    use_sh: bool = True,
) -> Tensor:  # (batch, 3, height, width)
    assert use_sh or gaussian_sh_coefficients.shape[-1] == 1
    viewbuf = _viewbuf(extrinsics, intrinsics, near, far, background_color, scale_invariant)
    _, _, _, n = gaussian_sh_coefficients.shape
    degree = isqrt(n) - 1
    # harmonics (b, g, 3, d_sh) and covariances (b, g, 3, 3) go to the operator as they are (no re-layout copies;
    # the reference permutes + gathers here, cuda_splatting.py:75,115,123)
    colors = gaussian_sh_coefficients if use_sh else gaussian_sh_coefficients[:, :, :, 0]
    color, _, _ = rasterize_views(
        gaussian_means, gaussian_covariances, gaussian_opacities, colors, viewbuf,
        image_shape=image_shape, sh_degree=degree, use_sh=use_sh, views_per_set=1, sh_planar=True, cov_3x3=True)
    return color


def render_cuda_orthographic(
    extrinsics: Tensor,  # (batch, 4, 4)
    width: Tensor,  # (batch,)
    height: Tensor,  # (batch,)
    near: Tensor,
    far: Tensor,
    image_shape: tuple,
    background_color: Tensor,  # (batch, 3)
    gaussian_means: Tensor,
    gaussian_covariances: Tensor,
    gaussian_sh_coefficients: Tensor,
    gaussian_opacities: Tensor,
    fov_degrees: float = 0.1,
    use_sh: bool = True,
    dump: Optional[dict] = None,
) -> Tensor:
    """Fake orthographic projection: camera moved back, tiny field of view (reference :130-220).
    Keeps the reference's `fov_y = atan(2 tan_fov_y)` quirk (:160)."""
    b, _, _ = extrinsics.shape
    assert use_sh or gaussian_sh_coefficients.shape[-1] == 1
    _, _, _, n = gaussian_sh_coefficients.shape
    degree = isqrt(n) - 1
    with torch.no_grad():
        fov_x = torch.tensor(fov_degrees, device=extrinsics.device).deg2rad()
        tan_fov_x = (0.5 * fov_x).tan()
        distance_to_near = (0.5 * width) / tan_fov_x
        tan_fov_y = 0.5 * height / distance_to_near
        fov_y = (2 * tan_fov_y).atan()
        near = near + distance_to_near
        far = far + distance_to_near
        move_back = torch.eye(4, dtype=torch.float32, device=extrinsics.device).repeat(b, 1, 1)
        move_back[:, 2, 3] = -distance_to_near
        extrinsics = extrinsics @ move_back
        if dump is not None:
            dump["extrinsics"] = extrinsics
            dump["fov_x"] = fov_x
            dump["fov_y"] = fov_y
            dump["near"] = near
            dump["far"] = far
        projection_matrix = get_projection_matrix(near, far, fov_x.expand(b), fov_y).transpose(-1, -2)
        view_matrix = extrinsics.inverse().transpose(-1, -2)
        full_projection = view_matrix @ projection_matrix
        viewbuf = pack_views(view_matrix, full_projection, extrinsics[:, :3, 3], tan_fov_x.expand(b),
                             tan_fov_y.expand(b) if tan_fov_y.dim() == 0 else tan_fov_y, background_color, None)
    # harmonics (b, g, 3, d_sh) and covariances (b, g, 3, 3) go to the operator as they are (as in render_cuda)
    colors = gaussian_sh_coefficients if use_sh else gaussian_sh_coefficients[:, :, :, 0]
    color, _, _ = rasterize_views(
        gaussian_means, gaussian_covariances, gaussian_opacities, colors, viewbuf,
        image_shape=image_shape, sh_degree=degree, use_sh=use_sh, views_per_set=1, sh_planar=True, cov_3x3=True)
    return color


def depth_fake_color(extrinsics: Tensor, gaussian_means: Tensor, near: Tensor, far: Tensor,
                     mode: DepthRenderingMode) -> Tensor:
    """Per-(view, Gaussian) scalar that the depth render blends (reference :238-251): camera-space z in
    un-normalised units, mapped by `mode` (the `log` mode keeps the reference's min/max quirk, :251)."""
    camera_space = torch.einsum("bij,bgj->bgi", extrinsics.inverse(), homogenize_points(gaussian_means))
    fake_color = camera_space[..., 2]
    if mode == "disparity":
        fake_color = 1 / fake_color
    elif mode == "relative_disparity":
        fake_color = depth_to_relative_disparity(fake_color, near[:, None], far[:, None])
    elif mode == "log":
        fake_color = fake_color.minimum(near[:, None]).maximum(far[:, None]).log()
    return fake_color


def render_depth_cuda(
    extrinsics: Tensor,
    intrinsics: Tensor,
    near: Tensor,
    far: Tensor,
    image_shape: tuple,
    gaussian_means: Tensor,
    gaussian_covariances: Tensor,
    gaussian_opacities: Tensor,
    scale_invariant: bool = True,
    mode: DepthRenderingMode = "depth",
) -> Tensor:  # (batch, height, width)
    """Depth image = sum_i f(z_i) alpha_i T_i (reference :226-269).  The reference renders f(z) as a
    3-channel precomputed colour and averages the channels; the three channels are identical, so it
    is blended here once, as the extra channel of a colour-less pass, with f(z) (`depth_fake_color`
    below states it in torch) evaluated inside the kernels.  Gaussians receive the same gradients as in
    the reference; the (unused) gradient the reference's torch graph sends to `extrinsics` through
    `extrinsics.inverse()` is not produced."""
    b, g = gaussian_opacities.shape
    dev = gaussian_means.device
    bg = torch.zeros((b, 3), dtype=torch.float32, device=dev)
    viewbuf = _viewbuf(extrinsics, intrinsics, near, far, bg, scale_invariant)
    zero_rgb = torch.zeros((b, g, 3), dtype=torch.float32, device=dev)
    _, depth, _ = rasterize_views(
        gaussian_means, gaussian_covariances, gaussian_opacities, zero_rgb, viewbuf,
        image_shape=image_shape, sh_degree=0, use_sh=False, views_per_set=1, extra_mode=mode, cov_3x3=True)
    return depth


def render_views(
    extrinsics: Tensor,  # (scene, view, 4, 4)
    intrinsics: Tensor,  # (scene, view, 3, 3)
    near: Tensor,  # (scene, view)
    far: Tensor,  # (scene, view)
    image_shape: tuple,
    background_color: Tensor,  # (3,)
    gaussian_means: Tensor,  # (scene, gaussian, 3)
    gaussian_covariances: Tensor,  # (scene, gaussian, 3, 3)
    gaussian_sh_coefficients: Tensor,  # (scene, gaussian, 3, d_sh)
    gaussian_opacities: Tensor,  # (scene, gaussian)
    depth_mode: Optional[DepthRenderingMode] = None,
    scale_invariant: bool = True,
):
    """Fused decoder path: all views of all scenes in one launch chain, Gaussians read once per scene
    (no V-fold `repeat`, reference decoder_splatting_cuda.py:52-56), depth as a 4th blended channel.
    Returns (color (scene, view, 3, h, w), depth (scene, view, h, w) | None)."""
    s, v = extrinsics.shape[:2]
    ext = extrinsics.reshape(s * v, 4, 4)
    intr = intrinsics.reshape(s * v, 3, 3)
    nr, fr = near.reshape(s * v), far.reshape(s * v)
    _, _, _, n = gaussian_sh_coefficients.shape
    degree = isqrt(n) - 1
    viewbuf = _viewbuf(ext, intr, nr, fr, background_color.reshape(3), scale_invariant)
    color, depth, _ = rasterize_views(
        gaussian_means, gaussian_covariances, gaussian_opacities, gaussian_sh_coefficients, viewbuf,
        image_shape=image_shape, sh_degree=degree, use_sh=True, views_per_set=v, extra_mode=depth_mode, sh_planar=True,
        cov_3x3=True)
    h, w = image_shape
    color = color.reshape(s, v, 3, h, w)
    if depth is not None:
        depth = depth.reshape(s, v, h, w)
    return color, depth
