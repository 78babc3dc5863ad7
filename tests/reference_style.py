"""The REFERENCE's way of calling the operator, restated so that tests and bench.py can drive the drop-in module
`diff_gaussian_rasterization` exactly as PF3plat's own decoder does, without the reference's source travelling:

  * `reference_style_render`  - what `render_cuda` does around the rasterizer (src/model/decoder/cuda_splatting.py:47-127):
    the 1 / near pre-scale as torch ops, the SH re-layout copy, fov / projection / inverse, then a PYTHON LOOP over the views
    with two `.item()` host syncs, a fresh 12-field settings object, a fresh `GaussianRasterizer`, a zero `means2D` leaf with
    grad and a fancy-index gather of the covariance's upper triangle per view;
  * `reference_style_decoder_forward` - what `DecoderSplattingCUDA.forward` does in front of it
    (src/model/decoder/decoder_splatting_cuda.py:44-67): flatten (b, v) and `repeat` every Gaussian tensor v times.

Test / measurement infrastructure: never imported by the product (`pf3plat_amd.render_cuda` is the fused counterpart)."""
from __future__ import annotations

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from pf3plat_amd.geometry import get_fov, get_projection_matrix


def reference_style_render(ext, intr, near, far, hw, bg, means, cov, sh, op, scale_invariant=True):
    """Per-view Python loop with explicit torch pre-scaling, exactly the reference's structure."""
    if scale_invariant:
        scale = 1 / near
        ext = ext.clone()
        ext[..., :3, 3] = ext[..., :3, 3] * scale[:, None]
        cov = cov * (scale[:, None, None, None] ** 2)
        means = means * scale[:, None, None]
        near, far = near * scale, far * scale
    shs = sh.permute(0, 1, 3, 2).contiguous()
    fov_x, fov_y = get_fov(intr).unbind(-1)
    tan_x, tan_y = (0.5 * fov_x).tan(), (0.5 * fov_y).tan()
    proj = get_projection_matrix(near, far, fov_x, fov_y).transpose(-1, -2)
    view = ext.inverse().transpose(-1, -2)
    full = view @ proj
    degree = int(round(sh.shape[-1] ** 0.5)) - 1
    imgs = []
    for i in range(ext.shape[0]):
        mean_gradients = torch.zeros_like(means[i], requires_grad=True)
        s = GaussianRasterizationSettings(hw[0], hw[1], tan_x[i].item(), tan_y[i].item(), bg[i], 1.0, view[i], full[i], degree,
                                          ext[i, :3, 3], False, False)
        row, col = torch.triu_indices(3, 3)
        img, _ = GaussianRasterizer(s)(means3D=means[i], means2D=mean_gradients, shs=shs[i], opacities=op[i, ..., None],
                                       cov3D_precomp=cov[i][:, row, col])
        imgs.append(img)
    return torch.stack(imgs)


def reference_style_decoder_forward(gaussians, extrinsics, intrinsics, near, far, hw, background_color):
    """(b, v) flattened, every Gaussian tensor repeated v times, then the per-view loop: -> (b, v, 3, h, w)."""
    b, v = extrinsics.shape[:2]
    rep = lambda t: t.repeat_interleave(v, dim=0)
    color = reference_style_render(
        extrinsics.reshape(b * v, 4, 4), intrinsics.reshape(b * v, 3, 3), near.reshape(b * v), far.reshape(b * v), hw,
        background_color.reshape(1, 3).expand(b * v, 3), rep(gaussians.means), rep(gaussians.covariances), rep(gaussians.harmonics),
        rep(gaussians.opacities))
    return color.reshape(b, v, 3, *hw)
