"""Host-logic tests on CPU: the wrappers' autograd chain (scale-invariant pre-scale applied inside the operator, cov 3x3 -> 6
gather, SH layout, fused multi-view sharing, depth as extra channel) against a literal per-view restatement of what the
reference wrapper does with torch ops around a per-view rasterizer (here: the oracle), src/model/decoder/cuda_splatting.py:47-127."""
import numpy as np
import pytest
import torch

import pf3plat_amd
from pf3plat_amd import synthetic
from pf3plat_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from pf3plat_amd.geometry import get_fov, get_projection_matrix
from pf3plat_amd.types import Gaussians
from tests.reference_style import reference_style_decoder_forward, reference_style_render
from tests.util import rel_l2


def _leaves(sc, b):
    g = sc.gaussians
    return [x.detach().clone().expand(b, *x.shape[1:]).contiguous().requires_grad_(True) for x in (g.means, g.covariances, g.harmonics, g.opacities)]


def test_render_cuda_equals_reference_style_loop_forward_and_backward(oracle_backend):
    sc = synthetic.make_scene(3, 300, (24, 32), num_views=2, near=2.0)
    bg = torch.tensor([[0.1, 0.2, 0.3], [0.0, 0.5, 1.0]])
    w = torch.rand((2, 3, 24, 32), generator=torch.Generator().manual_seed(0))
    a = _leaves(sc, 2)
    img_a = pf3plat_amd.render_cuda(sc.extrinsics[0], sc.intrinsics[0], sc.near[0], sc.far[0], (24, 32), bg, *a)
    (img_a * w).sum().backward()
    b = _leaves(sc, 2)
    img_b = reference_style_render(sc.extrinsics[0], sc.intrinsics[0], sc.near[0], sc.far[0], (24, 32), bg, *b)
    (img_b * w).sum().backward()
    assert rel_l2(img_a.detach().numpy(), img_b.detach().numpy()) < 1e-6
    for x, y, name in zip(a, b, ("means", "cov", "sh", "opac")):
        assert rel_l2(x.grad.numpy(), y.grad.numpy()) < 2e-5, name
    # gradient reaches only the 6 upper-triangle covariance entries (reference: fancy-index gather)
    assert torch.all(a[1].grad[..., 1, 0] == 0) and torch.all(a[1].grad[..., 2, 0] == 0) and torch.all(a[1].grad[..., 2, 1] == 0)
    assert a[1].grad[..., 0, 1].abs().sum() > 0


def test_fused_decoder_equals_repeat_then_render(oracle_backend):
    """DecoderSplattingCUDA.forward (shared Gaussians, one call) == reference structure (repeat V times, per-view render)."""
    sc = synthetic.make_scene(4, 200, (16, 24), num_views=3, near=1.5)
    dec = pf3plat_amd.DecoderSplattingCUDA(dataset_cfg=pf3plat_amd.decoder.DatasetCfgLike((0.2, 0.1, 0.0)))
    w = torch.rand((1, 3, 3, 16, 24), generator=torch.Generator().manual_seed(1))
    a = _leaves(sc, 1)
    out = dec.forward(Gaussians(*a), sc.extrinsics, sc.intrinsics, sc.near, sc.far, (16, 24), depth_mode=None)
    (out.color * w).sum().backward()
    assert out.depth is None
    b = _leaves(sc, 1)
    img = reference_style_decoder_forward(Gaussians(*b), sc.extrinsics, sc.intrinsics, sc.near, sc.far, (16, 24),
                                          torch.tensor([0.2, 0.1, 0.0]))
    (img * w).sum().backward()
    assert rel_l2(out.color.detach().numpy(), img.detach().numpy()) < 1e-6
    for x, y, name in zip(a, b, ("means", "cov", "sh", "opac")):
        assert rel_l2(x.grad.numpy(), y.grad.numpy()) < 2e-5, name


@pytest.mark.parametrize("mode", ["depth", "disparity", "relative_disparity"])
def test_depth_gradients_flow_through_fake_colour_and_blend(oracle_backend, mode):
    sc = synthetic.make_scene(5, 150, (16, 16), near=1.2)
    m, c, h, o = _leaves(sc, 1)
    d = pf3plat_amd.render_depth_cuda(sc.extrinsics[0], sc.intrinsics[0], sc.near[0], sc.far[0], (16, 16), m, c, o, mode=mode)
    d.sum().backward()
    assert d.shape == (1, 16, 16) and torch.isfinite(d).all()
    assert m.grad.abs().sum() > 0 and c.grad.abs().sum() > 0 and o.grad.abs().sum() > 0
    # finite-difference spot check on one mean coordinate (fp32 oracle => loose)
    with torch.no_grad():
        i = int(torch.argmax(m.grad[0, :, 2].abs()))
        eps = 1e-3
        mp, mm = m.detach().clone(), m.detach().clone()
        mp[0, i, 2] += eps
        mm[0, i, 2] -= eps
        f = lambda x: pf3plat_amd.render_depth_cuda(sc.extrinsics[0], sc.intrinsics[0], sc.near[0], sc.far[0], (16, 16), x,
                                                    c.detach(), o.detach(), mode=mode).sum().item()
        fd = (f(mp) - f(mm)) / (2 * eps)
    assert abs(fd - m.grad[0, i, 2].item()) < 0.05 * max(1.0, abs(fd))


def test_mark_visible_and_empty_scene(oracle_backend):
    s = GaussianRasterizationSettings(8, 8, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0, torch.zeros(3), False, False)
    r = GaussianRasterizer(s)
    vis = r.markVisible(torch.tensor([[0, 0, 1.0], [0, 0, 0.1], [0, 0, -1.0]]))
    assert vis.tolist() == [True, False, False]
    img, radii = r(means3D=torch.zeros(0, 3), means2D=torch.zeros(0, 3), opacities=torch.zeros(0, 1), colors_precomp=torch.zeros(0, 3),
                   cov3D_precomp=torch.zeros(0, 6))
    assert img.shape == (3, 8, 8) and radii.shape == (0,) and torch.all(img == 0)


@pytest.mark.parametrize("mode", ["depth", "disparity", "relative_disparity", "log"])
def test_builtin_depth_channel_equals_explicit_fake_colour(oracle_backend, mode):
    """extra_mode (f(z) evaluated by the operator, gradient folded into d_means) == the reference's formulation: f(z) as an
    explicit torch tensor (depth_fake_color, cuda_splatting.py:238-251) blended as an extra array, autograd doing the chain."""
    from pf3plat_amd.geometry import depth_to_relative_disparity, homogenize_points
    from pf3plat_amd.rasterizer import rasterize_views
    from pf3plat_amd.splatting import _viewbuf

    def depth_fake_color(extrinsics, means, near, far, mode):  # the scalar the reference blends, cuda_splatting.py:238-251
        z = torch.einsum("bij,bgj->bgi", extrinsics.inverse(), homogenize_points(means))[..., 2]
        if mode == "disparity":
            return 1 / z
        if mode == "relative_disparity":
            return depth_to_relative_disparity(z, near[:, None], far[:, None])
        if mode == "log":
            return z.minimum(near[:, None]).maximum(far[:, None]).log()
        return z

    sc = synthetic.make_scene(6, 200, (16, 20), num_views=2, near=1.7)
    ext, intr, nr, fr = sc.extrinsics[0], sc.intrinsics[0], sc.near[0], sc.far[0]
    w = torch.rand((2, 16, 20), generator=torch.Generator().manual_seed(2))
    outs = []
    for builtin in (True, False):
        m, c, h, o = _leaves(sc, 2)
        vb = _viewbuf(ext, intr, nr, fr, torch.zeros(3), True)
        kw = dict(image_shape=(16, 20), sh_degree=0, use_sh=False, views_per_set=1, cov_3x3=True)
        zero = torch.zeros((2, 200, 3))
        if builtin:
            _, d, _ = rasterize_views(m, c, o, zero, vb, extra_mode=mode, **kw)
        else:
            _, d, _ = rasterize_views(m, c, o, zero, vb, extra=depth_fake_color(ext, m, nr, fr, mode), **kw)
        (d * w).sum().backward()
        outs.append((d.detach().numpy(), m.grad.numpy(), c.grad.numpy(), o.grad.numpy()))
    for a, b, name in zip(outs[0], outs[1], ("depth", "d_means", "d_cov", "d_opac")):
        assert rel_l2(a, b) < 5e-5, (mode, name, rel_l2(a, b))


def test_render_depth_shares_one_gaussian_copy_between_the_views_of_a_scene(oracle_backend):
    """render_depth_cuda with (scenes x views) cameras and ONE set of Gaussians per scene == the reference's shape (Gaussians
    repeated per view), and the decoder's render_depth makes no per-view copies of the Gaussians."""
    sc = synthetic.make_scene(7, 150, (16, 16), num_views=3, near=1.3)
    g = sc.gaussians
    ext, intr, nr, fr = sc.extrinsics[0], sc.intrinsics[0], sc.near[0], sc.far[0]
    shared = pf3plat_amd.render_depth_cuda(ext, intr, nr, fr, (16, 16), g.means, g.covariances, g.opacities, mode="disparity")
    rep = lambda t: t.expand(3, *t.shape[1:]).contiguous()
    repeated = pf3plat_amd.render_depth_cuda(ext, intr, nr, fr, (16, 16), rep(g.means), rep(g.covariances), rep(g.opacities),
                                             mode="disparity")
    assert shared.shape == (3, 16, 16) and torch.equal(shared, repeated)
    oracle_backend.record = True
    dec = pf3plat_amd.DecoderSplattingCUDA()
    d = dec.render_depth(g, sc.extrinsics, sc.intrinsics, sc.near, sc.far, (16, 16), mode="disparity")
    assert torch.equal(d[0], shared) and len(oracle_backend.calls) == 1 and len(oracle_backend.calls[0]) == 3


@pytest.mark.parametrize("mode", ["depth", "relative_disparity"])
def test_depth_render_sends_the_references_gradient_to_extrinsics(oracle_backend, mode):
    """The reference's depth render reaches `extrinsics` through `extrinsics.inverse()` (cuda_splatting.py:239-242) - the one place a
    camera gets a gradient, and in training the extrinsics do require grad (model_wrapper.py:148-150).  Both entry points
    (`render_depth_cuda`, the fused `DecoderSplattingCUDA.forward(depth_mode=)`) must produce it, equal to a literal restatement
    of the reference's graph: f(z) as a torch tensor, rendered as a precomputed colour, channels averaged."""
    from pf3plat_amd.geometry import depth_to_relative_disparity, homogenize_points

    sc = synthetic.make_scene(8, 120, (16, 16), num_views=2, near=1.4)
    g = sc.gaussians
    w = torch.rand((2, 16, 16), generator=torch.Generator().manual_seed(3))
    ext0, intr, nr, fr = sc.extrinsics[0], sc.intrinsics[0], sc.near[0], sc.far[0]

    def leaves():
        return ext0.clone().requires_grad_(True), g.means.clone().requires_grad_(True)

    # (a) the reference's structure: Gaussians repeated per view, fake colour by torch ops, 3 equal channels, mean over them
    ext, means = leaves()
    rep = lambda t: t.expand(2, *t.shape[1:])
    z = torch.einsum("bij,bgj->bgi", ext.inverse(), homogenize_points(rep(means)))[..., 2]
    fake = depth_to_relative_disparity(z, nr[:, None], fr[:, None]) if mode == "relative_disparity" else z
    img = pf3plat_amd.render_cuda(ext.detach(), intr, nr, fr, (16, 16), torch.zeros((2, 3)), rep(means), rep(g.covariances),
                                  fake[..., None, None].expand(-1, -1, 3, 1), rep(g.opacities), use_sh=False).mean(dim=1)
    (img * w).sum().backward()
    want = (img.detach(), ext.grad.clone(), means.grad.clone())
    assert want[1][:, :3].abs().sum() > 0
    # (b) render_depth_cuda, Gaussians shared between the two views
    ext, means = leaves()
    d = pf3plat_amd.render_depth_cuda(ext, intr, nr, fr, (16, 16), means, g.covariances, g.opacities, mode=mode)
    (d * w).sum().backward()
    assert rel_l2(d.detach().numpy(), want[0].numpy()) < 1e-6
    assert rel_l2(ext.grad.numpy(), want[1].numpy()) < 2e-5 and rel_l2(means.grad.numpy(), want[2].numpy()) < 2e-5
    # (c) the decoder's fused colour + depth pass
    ext, means = leaves()
    out = pf3plat_amd.DecoderSplattingCUDA().forward(Gaussians(means, g.covariances, g.harmonics, g.opacities), ext[None], sc.intrinsics,
                                                     sc.near, sc.far, (16, 16), depth_mode=mode)
    (out.depth[0] * w).sum().backward()
    assert rel_l2(ext.grad.numpy(), want[1].numpy()) < 2e-5 and rel_l2(means.grad.numpy(), want[2].numpy()) < 2e-5
    # without a camera that requires grad the kernels evaluate f(z) themselves: same image, same Gaussian gradient
    means2 = g.means.clone().requires_grad_(True)
    d2 = pf3plat_amd.render_depth_cuda(ext0, intr, nr, fr, (16, 16), means2, g.covariances, g.opacities, mode=mode)
    (d2 * w).sum().backward()
    assert rel_l2(d2.detach().numpy(), want[0].numpy()) < 1e-6 and rel_l2(means2.grad.numpy(), want[2].numpy()) < 5e-5


def test_render_depth_takes_the_adapters_scale_rotation_form(oracle_backend):
    """Gaussians as the adapter emits them (covariances None; scales, quaternions, frames) through `render_depth` and
    `render_depth_cuda` == the same Gaussians with materialised covariances (this raised AttributeError)."""
    from oracle import adapter as oracle_adapter

    sc = synthetic.make_scene(9, 96, (16, 16), num_views=2, near=1.3)
    gen = torch.Generator().manual_seed(4)
    scales = 0.05 + 0.2 * torch.rand((1, 96, 3), generator=gen)
    quats = torch.randn((1, 96, 4), generator=gen)
    frames = torch.linalg.qr(torch.randn((1, 2, 3, 3), generator=gen))[0]
    cov = oracle_adapter.covariance_from_scale_rotation(torch.cat((scales, quats), -1), frames)
    g = sc.gaussians
    dec = pf3plat_amd.DecoderSplattingCUDA()
    a = (sc.extrinsics, sc.intrinsics, sc.near, sc.far, (16, 16))
    d_cov = dec.render_depth(Gaussians(g.means, cov, g.harmonics, g.opacities), *a, mode="disparity")
    adapted = Gaussians(g.means, None, g.harmonics, g.opacities, scales=scales, rotations=quats, frames=frames)
    d_sr = dec.render_depth(adapted, *a, mode="disparity")
    assert d_sr.shape == (1, 2, 16, 16) and rel_l2(d_sr.numpy(), d_cov.numpy()) < 1e-5
    c = adapted.clone()
    assert c.covariances is None and torch.equal(c.scales, scales) and c.scales is not scales and torch.equal(c.frames, frames)
    c2 = Gaussians(g.means, cov, g.harmonics, g.opacities).clone()
    assert c2.scales is None and torch.equal(c2.covariances, cov) and c2.covariances is not cov
