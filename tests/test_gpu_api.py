"""`-m gpu`: the Python operator surface on the device — the per-view `GaussianRasterizer` (upstream signature), the
reference-shaped wrappers `render_cuda` / `render_depth_cuda` / `render_cuda_orthographic`, and the fused
`DecoderSplattingCUDA.forward`, each against the same wrapper code driven by the oracle backend on CPU."""
import numpy as np
import pytest
import torch

import pf3plat_amd
from pf3plat_amd import rasterizer, synthetic
from pf3plat_amd.types import Gaussians
from tests.oracle_backend import OracleBackend
from tests.util import install_backend, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _with_oracle(fn):
    old = install_backend(OracleBackend(threads=8))
    try:
        return fn()
    finally:
        install_backend(old)


def _leafs(sc, device):
    g = sc.gaussians
    return [t.detach().clone().to(device).requires_grad_(True) for t in (g.means, g.covariances, g.harmonics, g.opacities)]


def _cmp_grads(a, b, tol=1e-4):
    for x, y, name in zip(a, b, ("means", "covariances", "harmonics", "opacities")):
        assert rel_l2(x.grad.cpu().numpy(), y.grad.cpu().numpy()) < tol, name


def test_library_is_loaded_from_tree_and_fails_loudly_on_cpu_tensors():
    be = rasterizer.get_backend()
    assert isinstance(be, rasterizer.HipBackend)
    assert "gfx950" in be.lib.gsr_build_info().decode()
    sc = synthetic.make_scene(1, 100, (16, 16))
    g = sc.gaussians
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pf3plat_amd.render_cuda(sc.extrinsics[0], sc.intrinsics[0], sc.near[0], sc.far[0], (16, 16), sc.background[None],
                                g.means, g.covariances, g.harmonics, g.opacities)


def test_render_cuda_forward_backward_matches_oracle_driven_wrapper():
    sc = synthetic.make_scene(31, 3000, (64, 64), num_views=2, near=2.0)
    b = 2
    w = torch.rand((b, 3, 64, 64), generator=torch.Generator().manual_seed(1))

    def run(device):
        m, c, h, o = _leafs(sc, device)
        rep = lambda t: t.expand(b, *t.shape[1:])
        img = pf3plat_amd.render_cuda(sc.extrinsics[0].to(device), sc.intrinsics[0].to(device), sc.near[0].to(device),
                                      sc.far[0].to(device), (64, 64), torch.tensor([[0.1, 0.2, 0.3]] * b, device=device),
                                      rep(m), rep(c), rep(h), rep(o))
        (img * w.to(device)).sum().backward()
        return img.detach().cpu().numpy(), (m, c, h, o)

    gi, gl = run(DEV)
    oi, ol = _with_oracle(lambda: run("cpu"))
    assert rel_l2(gi, oi) < 1e-4
    _cmp_grads(gl, ol)
    # covariance gradient lands only on the upper triangle, like the reference's fancy-index gather
    assert torch.all(gl[1].grad[..., 1, 0] == 0) and torch.all(gl[1].grad[..., 2, 0] == 0) and torch.all(gl[1].grad[..., 2, 1] == 0)


@pytest.mark.parametrize("mode", ["depth", "disparity", "relative_disparity", "log"])
def test_render_depth_cuda_modes(mode):
    sc = synthetic.make_scene(32, 2000, (48, 48), near=1.5)

    def run(device):
        m, c, h, o = _leafs(sc, device)
        d = pf3plat_amd.render_depth_cuda(sc.extrinsics[0].to(device), sc.intrinsics[0].to(device), sc.near[0].to(device),
                                          sc.far[0].to(device), (48, 48), m, c, o, mode=mode)
        d.sum().backward()
        return d.detach().cpu().numpy(), (m, c, o)

    gd, gl = run(DEV)
    od, ol = _with_oracle(lambda: run("cpu"))
    assert gd.shape == (1, 48, 48) and rel_l2(gd, od) < 1e-4
    for x, y in zip(gl, ol):
        assert rel_l2(x.grad.cpu().numpy(), y.grad.cpu().numpy()) < 1e-4


def test_render_cuda_orthographic():
    sc = synthetic.make_scene(33, 2000, (48, 48))
    g = sc.gaussians
    ext = torch.eye(4)[None].clone()
    ext[0, 2, 3] = -2.0

    def run(device):
        dump = {}
        t = lambda x: x.to(device)
        img = pf3plat_amd.render_cuda_orthographic(t(ext), torch.tensor([6.0], device=device), torch.tensor([6.0], device=device),
                                                   torch.tensor([0.0], device=device), torch.tensor([40.0], device=device), (48, 48),
                                                   torch.zeros((1, 3), device=device), t(g.means), t(g.covariances), t(g.harmonics),
                                                   t(g.opacities), fov_degrees=10.0, dump=dump)
        assert set(dump) == {"extrinsics", "fov_x", "fov_y", "near", "far"}
        return img.cpu().numpy()

    gi = run(DEV)
    oi = _with_oracle(lambda: run("cpu"))
    assert gi.max() > 0.05 and rel_l2(gi, oi) < 1e-4


def test_decoder_forward_fused_colour_and_depth_config4_shape():
    """BASELINE configs[3] shape at reduced G for the oracle's sake: B=1, V=3 target views, colour + depth in one pass."""
    sc = synthetic.make_scene(50, 20000, (128, 128), num_views=3, near=1.2)
    dec_gpu = pf3plat_amd.DecoderSplattingCUDA().to(DEV)
    dec_cpu = pf3plat_amd.DecoderSplattingCUDA()
    w = torch.rand((1, 3, 3, 128, 128), generator=torch.Generator().manual_seed(2))
    wd = torch.rand((1, 3, 128, 128), generator=torch.Generator().manual_seed(3)) * 0.1

    def run(dec, device):
        m, c, h, o = _leafs(sc, device)
        out = dec.forward(Gaussians(m, c, h, o), sc.extrinsics.to(device), sc.intrinsics.to(device), sc.near.to(device),
                          sc.far.to(device), (128, 128), depth_mode="depth")
        ((out.color * w.to(device)).sum() + (out.depth * wd.to(device)).sum()).backward()
        return out.color.detach().cpu().numpy(), out.depth.detach().cpu().numpy(), (m, c, h, o)

    gc, gd, gl = run(dec_gpu, DEV)
    oc, od, ol = _with_oracle(lambda: run(dec_cpu, "cpu"))
    assert gc.shape == (1, 3, 3, 128, 128) and gd.shape == (1, 3, 128, 128)
    assert rel_l2(gc, oc) < 1e-4 and rel_l2(gd, od) < 1e-4
    _cmp_grads(gl, ol)
    # and the separate depth pass agrees with the fused channel
    with torch.no_grad():
        d2 = dec_gpu.render_depth(Gaussians(*[t.detach() for t in gl]), sc.extrinsics.to(DEV), sc.intrinsics.to(DEV),
                                  sc.near.to(DEV), sc.far.to(DEV), (128, 128), mode="depth")
    assert rel_l2(d2.cpu().numpy(), gd) < 1e-5


def test_decoder_forward_config4_full_size_colour_and_depth_fwd_bwd():
    """BASELINE configs[3] as the reference runs it: `DecoderSplattingCUDA.forward` on B = 1, G = 131 072, K = 25, V = 3 target
    views, 256 x 256, colour + depth, forward and backward, against the same decoder code driven by the oracle on CPU
    (reference src/model/decoder/decoder_splatting_cuda.py:35-67).  Bounds hold over ALL elements (nothing set aside)."""
    sc = synthetic.make_scene(50, 131072, (256, 256), num_views=3)
    dec_gpu = pf3plat_amd.DecoderSplattingCUDA().to(DEV)
    dec_cpu = pf3plat_amd.DecoderSplattingCUDA()
    w = torch.rand((1, 3, 3, 256, 256), generator=torch.Generator().manual_seed(2))
    wd = torch.rand((1, 3, 256, 256), generator=torch.Generator().manual_seed(3)) * 0.1

    def run(dec, device):
        m, c, h, o = _leafs(sc, device)
        out = dec.forward(Gaussians(m, c, h, o), sc.extrinsics.to(device), sc.intrinsics.to(device), sc.near.to(device),
                          sc.far.to(device), (256, 256), depth_mode="depth")
        ((out.color * w.to(device)).sum() + (out.depth * wd.to(device)).sum()).backward()
        return out.color.detach().cpu().numpy(), out.depth.detach().cpu().numpy(), (m, c, h, o)

    gc, gd, gl = run(dec_gpu, DEV)
    oc, od, ol = _with_oracle(lambda: run(dec_cpu, "cpu"))
    assert gc.shape == (1, 3, 3, 256, 256) and gd.shape == (1, 3, 256, 256)
    assert rel_l2(gc, oc) < 1e-4 and rel_l2(gd, od) < 1e-4
    assert int((np.abs(gc - oc).max(axis=2) > 1e-4).sum()) <= 4  # pixels whose colour differs by more than 1e-4
    _cmp_grads(gl, ol)


@pytest.mark.parametrize("mode", ["depth", "relative_disparity"])
def test_reference_graph_training_call_config4_fused_depth_term_reaches_extrinsics(mode):
    """PF3plat's training call as the reference makes it: `extrinsics` requires grad (src/model/model_wrapper.py:148-156) and depth
    is rendered every step (config/main.yaml:50).  The reference's graph then sends ONE gradient to the camera - through the depth
    render's `extrinsics.inverse()` (cuda_splatting.py:239-251).  Fused path on the device (f(z) inside the kernels, the depth
    term of the camera gradient out of the backward preprocess, carried to `extrinsics` by the set-up's closed-form backward)
    against torch autograd over the literal graph - `depth_fake_color` blended as an explicit channel - on the oracle:
    BASELINE configs[3] size (B = 1, G = 131 072, V = 3, colour + depth), every Gaussian gradient and `extrinsics.grad`."""
    from pf3plat_amd.rasterizer import rasterize_views
    from pf3plat_amd.splatting import _viewbuf, depth_fake_color

    hw = (256, 256)
    sc = synthetic.make_scene(50, 131072, hw, num_views=3)
    w = torch.rand((1, 3, 3, *hw), generator=torch.Generator().manual_seed(2))
    wd = torch.rand((1, 3, *hw), generator=torch.Generator().manual_seed(3)) * 0.1

    def fused(device):
        m, c, h, o = _leafs(sc, device)
        ext = sc.extrinsics.clone().to(device).requires_grad_(True)
        dec = pf3plat_amd.DecoderSplattingCUDA().to(device)
        out = dec.forward(Gaussians(m, c, h, o), ext, sc.intrinsics.to(device), sc.near.to(device), sc.far.to(device), hw, depth_mode=mode)
        ((out.color * w.to(device)).sum() + (out.depth * wd.to(device)).sum()).backward()
        return out.color.detach().cpu().numpy(), out.depth.detach().cpu().numpy(), (m, c, h, o), ext.grad.cpu().numpy()

    def literal():  # (oracle backend, CPU): the reference's graph, f(z) by torch ops, autograd doing the chain to means AND extrinsics
        m, c, h, o = _leafs(sc, "cpu")
        ext = sc.extrinsics.clone().requires_grad_(True)
        e3, nr, fr = ext.reshape(3, 4, 4), sc.near.reshape(3), sc.far.reshape(3)
        vb = _viewbuf(e3.detach(), sc.intrinsics.reshape(3, 3, 3), nr, fr, torch.zeros(3), True)
        color, depth, _ = rasterize_views(m, c, o, h, vb, image_shape=hw, sh_degree=4, use_sh=True, views_per_set=3, sh_planar=True,
                                          cov_3x3=True, extra=depth_fake_color(e3, nr, fr, m, mode))
        ((color.reshape(1, 3, 3, *hw) * w).sum() + (depth.reshape(1, 3, *hw) * wd).sum()).backward()
        return color.detach().numpy().reshape(1, 3, 3, *hw), depth.detach().numpy().reshape(1, 3, *hw), (m, c, h, o), ext.grad.numpy()

    gc, gd, gl, ge = fused(DEV)
    oc, od, ol, oe = _with_oracle(literal)
    assert rel_l2(gc, oc) < 1e-4 and rel_l2(gd, od) < 1e-4
    _cmp_grads(gl, ol)
    assert np.abs(oe[0, :, :3]).sum() > 0 and rel_l2(ge, oe) < 1e-4, rel_l2(ge, oe)


def test_backward_twice_with_retain_graph_and_settings_debug():
    """Upstream's Function can be differentiated repeatedly (retain_graph=True / several autograd.grad calls over one render);
    `settings.debug=True` reaches the library (per-stage synchronise + check) and changes no result."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from tests.util import make_camera, random_small_scene

    sc = random_small_scene(5, 400, sh_coeffs=25, dtype=np.float32)
    cam = make_camera(dtype=np.float32)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=DEV)
    means = t(sc["means"]).requires_grad_(True)
    shs = t(sc["colors"]).requires_grad_(True)
    imgs = []
    for debug in (False, True):
        settings = GaussianRasterizationSettings(
            image_height=40, image_width=48, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=t([0.1, 0.1, 0.1]),
            scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]).reshape(4, 4), projmatrix=t(cam["projmatrix"]).reshape(4, 4),
            sh_degree=4, campos=t(cam["campos"]), prefiltered=False, debug=debug)
        img, _ = GaussianRasterizer(settings)(means3D=means, means2D=None, shs=shs, opacities=t(sc["opac"])[:, None],
                                              cov3D_precomp=t(sc["cov6"]))
        imgs.append(img)
    assert torch.equal(imgs[0], imgs[1])
    loss = imgs[0].sum()
    g1 = torch.autograd.grad(loss, [means, shs], retain_graph=True)
    g2 = torch.autograd.grad(loss, [means, shs], retain_graph=True)
    g3 = torch.autograd.grad(imgs[0][0].sum(), [means], retain_graph=False)
    for a, b in zip(g1, g2):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)
    assert torch.isfinite(g3[0]).all() and g3[0].abs().sum() > 0


def test_per_view_rasterizer_upstream_signature():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from tests.util import make_camera, random_small_scene

    sc = random_small_scene(2, 300, sh_coeffs=25, dtype=np.float32)
    cam = make_camera(dtype=np.float32)

    def run(device):
        t = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
        means = t(sc["means"]).requires_grad_(True)
        means2d = torch.zeros_like(means, requires_grad=True)
        shs = t(sc["colors"]).requires_grad_(True)
        settings = GaussianRasterizationSettings(
            image_height=40, image_width=48, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=t([0.1, 0.1, 0.1]),
            scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]).reshape(4, 4), projmatrix=t(cam["projmatrix"]).reshape(4, 4),
            sh_degree=4, campos=t(cam["campos"]), prefiltered=False, debug=False)
        rast = GaussianRasterizer(settings)
        img, radii = rast(means3D=means, means2D=means2d, shs=shs, colors_precomp=None, opacities=t(sc["opac"])[:, None],
                          cov3D_precomp=t(sc["cov6"]))
        assert img.shape == (3, 40, 48) and radii.shape == (300,) and radii.dtype == torch.int32
        img.sum().backward()
        vis = rast.markVisible(means.detach())
        with pytest.raises(Exception, match="excatly one"):
            rast(means3D=means, means2D=means2d, opacities=t(sc["opac"]), cov3D_precomp=t(sc["cov6"]))
        with pytest.raises(Exception, match="exactly one"):
            rast(means3D=means, means2D=means2d, shs=shs, opacities=t(sc["opac"]))
        # scales/rotations instead of cov3D_precomp (HIP: gsr_cov_from_scale_rot[_backward]; CPU side of the comparison:
        # the same arithmetic as torch ops in front of the oracle)
        scales = t(sc["scales"]).requires_grad_(True)
        rots = t(sc["rots"]).requires_grad_(True)
        img2, _ = rast(means3D=means, means2D=means2d, shs=shs, opacities=t(sc["opac"]), scales=scales, rotations=rots)
        wimg = torch.linspace(0.2, 1.0, img2.numel(), device=device).reshape(img2.shape)
        (img2 * wimg).sum().backward()
        settings15 = settings._replace(scale_modifier=1.5)
        img3, _ = GaussianRasterizer(settings15)(means3D=means, means2D=means2d, shs=shs, opacities=t(sc["opac"]), scales=scales,
                                                 rotations=rots)
        return (img.detach().cpu().numpy(), radii.cpu().numpy(), means.grad.cpu().numpy(), means2d.grad.cpu().numpy(),
                shs.grad.cpu().numpy(), vis.cpu().numpy(), img2.detach().cpu().numpy(), scales.grad.cpu().numpy(),
                rots.grad.cpu().numpy(), img3.detach().cpu().numpy())

    g = run(DEV)
    o = _with_oracle(lambda: run("cpu"))
    assert rel_l2(g[0], o[0]) < 1e-4 and np.array_equal(g[1], o[1])
    for i in (2, 3, 4):
        assert rel_l2(g[i], o[i]) < 1e-4, i
    assert np.array_equal(g[5], o[5]) and g[5].all()
    assert rel_l2(g[6], g[0]) < 1e-4  # scale/rotation path reproduces the precomputed covariance image
    assert rel_l2(g[7], o[7]) < 1e-4 and rel_l2(g[8], o[8]) < 1e-4  # dL/dscales, dL/drotations
    assert np.abs(o[7]).max() > 0 and np.abs(o[8]).max() > 0
    assert rel_l2(g[9], o[9]) < 1e-4 and rel_l2(g[9], g[6]) > 1e-2  # scale_modifier reaches the covariance


def test_setup_views_kernels_match_the_oracle_camera_arithmetic():
    """gsr_setup_views / gsr_setup_views_orthographic (one launch each) == oracle/cameras.py (which tests/test_wrapper_fixtures.py
    pins to the reference's cuda_splatting.py:64-87 and :153-181), for rotated/translated cameras, off-centre principal
    points, both scale modes, and the orthographic move-back with its dump values."""
    from oracle import cameras
    from tests.util import look_at_c2w

    v = 5
    gen = torch.Generator().manual_seed(3)
    ext = torch.stack([torch.tensor(look_at_c2w((0.3 * i - 0.5, 0.2 * i, -0.4 * i))) for i in range(v)])
    intr = torch.eye(3).repeat(v, 1, 1)
    intr[:, 0, 0] = 0.6 + 0.5 * torch.rand(v, generator=gen)
    intr[:, 1, 1] = 0.6 + 0.5 * torch.rand(v, generator=gen)
    intr[:, 0, 2] = 0.45 + 0.1 * torch.rand(v, generator=gen)
    intr[:, 1, 2] = 0.5
    near = 0.5 + 2 * torch.rand(v, generator=gen)
    far = 50 + 50 * torch.rand(v, generator=gen)
    bg = torch.rand(v, 3, generator=gen)
    be = rasterizer.get_backend()
    for si in (True, False):
        got = be.setup_views(ext.to(DEV), intr.to(DEV), near.to(DEV), far.to(DEV), bg.to(DEV), si).cpu()
        want = cameras.view_records(ext.numpy(), intr.numpy(), near.numpy(), far.numpy(), bg.numpy(), si)
        np.testing.assert_allclose(got.numpy(), want, rtol=2e-5, atol=2e-6)
    one_bg = be.setup_views(ext.to(DEV), intr.to(DEV), near.to(DEV), far.to(DEV), bg[0].to(DEV), True).cpu()
    assert torch.allclose(one_bg[:, 37:40], bg[0].expand(v, 3))
    width, height = 4 + 4 * torch.rand(v, generator=gen), 3 + 4 * torch.rand(v, generator=gen)
    for fov in (10.0, 0.1):
        got, dump = be.setup_views_orthographic(ext.to(DEV), width.to(DEV), height.to(DEV), near.to(DEV), far.to(DEV), bg.to(DEV), fov)
        want, wdump = cameras.view_records_orthographic(ext.numpy(), width.numpy(), height.numpy(), near.numpy(), far.numpy(),
                                                        bg.numpy(), fov)
        # the tiny field of view makes distances ~1e3..1e5: compare relative to the size of each record's entries
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * np.abs(want).max())
        for k in ("extrinsics", "fov_x", "fov_y", "near", "far"):
            np.testing.assert_allclose(dump[k].cpu().numpy(), wdump[k], rtol=1e-5, atol=1e-6, err_msg=k)


def test_reference_unchanged_decoder_path_at_config4_size():
    """BASELINE configs[3] at its real size (G = 131 072, 3 target views) rendered EXACTLY as the reference's decoder does it
    (tests/reference_style.py: V-fold `repeat`, torch pre-scale, per-view settings object with `.item()` floats, a fresh
    `GaussianRasterizer` per view through the `diff_gaussian_rasterization` module name, strided `campos`, default `sync`
    policy): HIP against the same Python driven by the oracle, forward and backward, and against the fused decoder."""
    from tests.reference_style import reference_style_decoder_forward

    n, hw = 131072, (256, 256)
    sc = synthetic.make_scene(50, n, hw, num_views=3)
    w = torch.rand((1, 3, 3, *hw), generator=torch.Generator().manual_seed(4))
    bgc = torch.tensor([0.1, 0.2, 0.3])

    def run(device):
        leaves = _leafs(sc, device)
        t = lambda x: x.to(device)
        img = reference_style_decoder_forward(Gaussians(*leaves), t(sc.extrinsics), t(sc.intrinsics), t(sc.near), t(sc.far), hw, t(bgc))
        (img * t(w)).sum().backward()
        return img.detach().cpu().numpy(), leaves

    assert rasterizer.get_backend().sync_policy == "sync"
    gi, gl = run(DEV)
    oi, ol = _with_oracle(lambda: run("cpu"))
    assert gi.shape == (1, 3, 3, *hw) and rel_l2(gi, oi) < 1e-4
    _cmp_grads(gl, ol)
    # and the fused decoder (one launch chain, no repeat) renders the same three views
    fl = _leafs(sc, DEV)
    dec = pf3plat_amd.DecoderSplattingCUDA(dataset_cfg=pf3plat_amd.decoder.DatasetCfgLike(tuple(bgc.tolist()))).to(DEV)
    out = dec.forward(Gaussians(*fl), sc.extrinsics.to(DEV), sc.intrinsics.to(DEV), sc.near.to(DEV), sc.far.to(DEV), hw)
    (out.color * w.to(DEV)).sum().backward()
    assert rel_l2(out.color.detach().cpu().numpy(), gi) < 2e-6
    _cmp_grads(fl, gl, tol=2e-5)


def test_reference_training_loop_skips_a_step_whose_pair_count_jumps_and_goes_on():
    """Ten training steps through the reference's own call structure (tests/reference_style.py: per-view operator, fresh
    settings + rasterizer + `means2D` leaf per view) with the reference's optimizer step around them (model_wrapper.py:210-241: a
    step whose gradients hold a NaN is skipped) - the trainer has opted into `on_overflow = "nan"`, as a loop with that guard does.
    The footprints change from step to step, as a scale head's output does in training.  Steps 0-3 read their status; from step 4
    on the forward is deferred and sized from the shape's history: running maximum x max(1.25, 1 + 4 sigma / mean).  Step 6 lists
    ~1.6x the pairs of step 5 - inside the head-room the history has earned: NOT skipped (round 5 sized at a fixed 1.25x and lost
    such a step).  At step 7 the footprints grow 36-fold (far more than 3x the pairs): NO exception - a warning, a NaN image, all-NaN
    gradients, the guard skips the step - and steps 8-9 (same large scene) are finite; step 9 matches the oracle-driven same code."""
    import warnings

    from pf3plat_amd._lib import RasterOverflowWarning
    from tests.reference_style import reference_style_decoder_forward

    n, hw = 6000, (64, 64)
    sc = synthetic.make_scene(61, n, hw, num_views=1)
    w = torch.rand((1, 1, 3, *hw), generator=torch.Generator().manual_seed(6))
    bgc = torch.tensor([0.1, 0.2, 0.3])
    key = (1, n, *hw)

    # what covariance factor lists 1.6x the pairs of factor 0.9?  (blocking calls of the plan API on a backend of its own)
    probe = rasterizer.HipBackend()
    ins = tuple(t.to(DEV).contiguous() for t in synthetic.scene_operator_inputs(sc))
    vb = synthetic.scene_viewbuf(sc).to(DEV)
    cfg = rasterizer.RasterConfig(1, 1, 1, n, *hw, 4, 25, 4, False)
    plan = probe.make_plan(cfg, torch.device(DEV), capacity=64 * n)

    def pairs_of(g):
        probe.run_forward(plan, vb, ins[0], ins[1] * g, ins[2], ins[3])
        st = probe.read_status(plan)
        assert not st["overflow"]
        return st["num_pairs"]

    base5 = pairs_of(0.9)
    g6 = min((1.5 + 0.25 * k for k in range(40)), key=lambda g: abs(pairs_of(g) / base5 - 1.6))
    assert 1.5 < pairs_of(g6) / base5 < 1.75, (g6, pairs_of(g6) / base5)
    grow = [1.0, 1.3, 0.8, 1.15, 1.0, 0.9, g6, 36.0, 36.0, 36.0]

    be = rasterizer.HipBackend()
    old = install_backend(be)
    try:
        assert be.sync_policy == "sync" and be.defer_after == 4 and be.on_overflow == "raise"  # the library's defaults
        be.on_overflow = "nan"  # this loop has the reference's NaN guard: it opts in
        leaves = _leafs(sc, DEV)
        opt = torch.optim.SGD(leaves, lr=1e-4)

        def step(k, leaves_, device):
            m, c, h, o = leaves_
            tt = lambda x: x.to(device)
            img = reference_style_decoder_forward(Gaussians(m, c * grow[k], h, o), tt(sc.extrinsics), tt(sc.intrinsics), tt(sc.near),
                                                  tt(sc.far), hw, tt(bgc))
            (img * tt(w)).sum().backward()
            return img.detach()

        pairs = []
        for k in range(10):
            before = [x.detach().clone() for x in leaves]
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                img = step(k, leaves, DEV)  # (no exception at any step)
            pairs.append(be.last_status["num_pairs"])
            nan_grad = any(torch.isnan(x.grad).any().item() for x in leaves)  # the reference's guard
            if k == 7:
                assert any(issubclass(c.category, RasterOverflowWarning) and "returns NaN gradients" in str(c.message) for c in caught)
                # (every gradient the operator hands out is NaN; the covariance leaf's lower triangle receives none at all - the
                # reference's triu gather - and stays 0)
                assert torch.isnan(img).all() and nan_grad and all(torch.isnan(x.grad).any() for x in leaves)
                assert all(torch.isnan(leaves[j].grad).all() for j in (0, 2, 3))
            else:
                assert not caught and torch.isfinite(img).all() and not nan_grad, (k, pairs)
            if k == 5:
                assert be.headroom_for(key) > 1.25  # the history's spread has widened the factor beyond the floor
            if k == 9:  # against the oracle-driven same code on the same parameter values
                cpu_leaves = [x.detach().clone().cpu().requires_grad_(True) for x in leaves]
                oi = _with_oracle(lambda: step(9, cpu_leaves, "cpu"))
                assert rel_l2(img.cpu().numpy(), oi.numpy()) < 1e-4
                _cmp_grads(leaves, cpu_leaves)
            if not nan_grad:
                opt.step()
            else:
                assert all(torch.equal(a, b.detach()) for a, b in zip(before, leaves))
            opt.zero_grad()
        assert be.seen[key] == 10 and not be.pending and len(be.poisoned) == 1
        assert pairs[6] > 1.45 * pairs[5], pairs  # the 1.6x jump was a real one - and was rendered, not skipped
        assert pairs[9] > 3 * pairs[6], pairs  # (pairs[7] is the skipped step's own count, read at its backward)
    finally:
        install_backend(old)


def test_pack_view_matches_the_torch_assembly_of_the_record():
    """gsr_pack_view (one launch per settings object) == pack_views (the torch assembly): strided campos, float and tensor tan-fovs."""
    from pf3plat_amd.rasterizer import GaussianRasterizationSettings, pack_views

    g = torch.Generator().manual_seed(3)
    ext = torch.randn((4, 4), generator=g).to(DEV)
    vm, pm, bg = torch.randn((4, 4), generator=g).to(DEV), torch.randn((4, 4), generator=g).to(DEV), torch.rand(3, generator=g).to(DEV)
    campos = ext[:3, 3]
    assert campos.stride(0) == 4
    be = rasterizer.get_backend()
    for tx, ty in ((0.61, 0.47), (torch.tensor([0.61], device=DEV), torch.tensor(0.47, device=DEV))):
        rs = GaussianRasterizationSettings(32, 48, tx, ty, bg, 1.5, vm, pm, 3, campos, False, False)
        rec = be.pack_view(rs, torch.device(DEV))
        f = lambda x: x.reshape(-1)[:1].float() if torch.is_tensor(x) else torch.tensor([x], device=DEV)
        want = pack_views(vm[None], pm[None], campos[None], f(tx), f(ty), bg[None], None, 1.5)
        assert rec.shape == (1, 48) and torch.equal(rec, want)


def test_two_host_threads_share_the_process_wide_backend():
    """The compiled backend releases the GIL around its work and keeps its bookkeeping (capacity hints, pending status copies, caches)
    under a lock: two host threads rendering different shapes through the process-wide backend, each on a stream of its own, get the
    images a single thread gets - under the default policy (blocking status reads) and under `lazy` (pending copies verified by
    whichever thread calls next)."""
    import threading

    be = rasterizer.HipBackend()
    old = install_backend(be)
    try:
        scenes = [synthetic.make_scene(71, 20000, (96, 96), num_views=2).to(DEV), synthetic.make_scene(72, 9000, (64, 128), num_views=1).to(DEV)]
        dec = pf3plat_amd.DecoderSplattingCUDA().to(DEV)

        def render(sc):
            with torch.no_grad():
                return dec.forward(sc.gaussians, sc.extrinsics, sc.intrinsics, sc.near, sc.far, tuple(sc.image_shape)).color

        want = [render(sc).clone() for sc in scenes]
        for policy in ("sync", "lazy"):
            be.sync_policy = policy
            got, errors = [[], []], []

            def worker(k):
                try:
                    stream = torch.cuda.Stream()
                    with torch.cuda.stream(stream):
                        for _ in range(25):
                            got[k].append(render(scenes[k]))
                    stream.synchronize()
                except Exception as e:  # pragma: no cover - reported below
                    errors.append(e)

            threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            be.check_pending(wait=True)
            assert not errors, errors
            for k in range(2):
                assert len(got[k]) == 25 and all(torch.equal(g, want[k]) for g in got[k]), (policy, k)
        assert not be.pending
    finally:
        install_backend(old)


@pytest.mark.parametrize("case", ["A", "B", "C", "D_depth", "D_disparity", "D_relative_disparity", "D_log", "E", "F"])
def test_reference_wrapper_fixture_cases_through_the_product_path(case):
    """tests/golden/wrapper_fixtures.npz (the inputs the REFERENCE's host wrapper was recorded on, tests/golden/make_wrapper_fixtures.py)
    through the PRODUCT path - compiled `rasterize_views` (csrc/gsr_torch.cpp: `prepare_call` + the operator) on the HIP library -
    against the same wrapper calls on the oracle backend, whose per-view arguments `tests/test_wrapper_fixtures.py` pins to the
    reference's recorded calls on CPU.  Both paths state the call shape through the same compiled `prepare_call`."""
    import os

    fix = np.load(os.path.join(os.path.dirname(__file__), "golden", "wrapper_fixtures.npz"))

    def run(device):
        t = lambda name: torch.tensor(fix[name]).to(device)
        k = case[0]
        hw = tuple(int(x) for x in fix[f"{k}_in_hw"])
        cams = (t(f"{k}_in_ext"), t(f"{k}_in_intr"), t(f"{k}_in_near"), t(f"{k}_in_far")) if k != "E" else None
        if k in "AB":
            return [pf3plat_amd.render_cuda(*cams, hw, t(f"{k}_in_bg"), t(f"{k}_in_means"), t(f"{k}_in_cov"), t(f"{k}_in_sh"), t(f"{k}_in_op"))]
        if k == "C":
            return [pf3plat_amd.render_cuda(*cams, hw, t("C_in_bg"), t("C_in_means"), t("C_in_cov"), t("C_in_sh"), t("C_in_op"),
                                            scale_invariant=False, use_sh=False)]
        if k == "D":
            return [pf3plat_amd.render_depth_cuda(*cams, hw, t("D_in_means"), t("D_in_cov"), t("D_in_op"), mode=case[2:])]
        if k == "E":
            return [pf3plat_amd.render_cuda_orthographic(t("E_in_ext"), t("E_in_width"), t("E_in_height"), t("E_in_near"), t("E_in_far"), hw,
                                                         t("E_in_bg"), t("E_in_means"), t("E_in_cov"), t("E_in_sh"), t("E_in_op"), fov_degrees=10.0)]
        dec = pf3plat_amd.DecoderSplattingCUDA(dataset_cfg=pf3plat_amd.decoder.DatasetCfgLike(tuple(fix["F_in_bgcolor"])), on_overflow=None).to(device)
        out = dec.forward(Gaussians(t("F_in_means"), t("F_in_cov"), t("F_in_sh"), t("F_in_op")), *cams, hw, depth_mode="depth")
        return [out.color, out.depth]

    assert isinstance(rasterizer.get_backend(), rasterizer.HipBackend)
    got = run(DEV)
    want = _with_oracle(lambda: run("cpu"))
    for a, b in zip(got, want):
        assert a.shape == b.shape and a.is_cuda
        a, b = a.cpu().numpy(), b.numpy()
        assert rel_l2(a, b) < 1e-4 or np.abs(a - b).max() < 1e-6, (case, rel_l2(a, b))


def test_per_view_rasterizer_under_stream_capture_returns_the_eager_shapes_and_module_hooks_run():
    """(i) `GaussianRasterizer` under HIP-graph capture returns (3, H, W) and (N) - what it returns eagerly (round 5's capture branch
    handed out the (1, 3, H, W) / (1, N) views) - and the replay reproduces the eager image; (ii) the module's `__call__` skips
    nn.Module's hook machinery only while no hook can be waiting: a registered forward hook (or a global module hook) runs."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from tests.util import make_camera, random_small_scene

    sc = random_small_scene(5, 400, sh_coeffs=25, dtype=np.float32)
    cam = make_camera(dtype=np.float32)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=DEV)
    settings = GaussianRasterizationSettings(
        image_height=40, image_width=48, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=t([0.1, 0.1, 0.1]), scale_modifier=1.0,
        viewmatrix=t(cam["viewmatrix"]).reshape(4, 4), projmatrix=t(cam["projmatrix"]).reshape(4, 4), sh_degree=4, campos=t(cam["campos"]),
        prefiltered=False, debug=False)
    args = dict(means3D=t(sc["means"]), means2D=None, shs=t(sc["colors"]), colors_precomp=None, opacities=t(sc["opac"])[:, None], cov3D_precomp=t(sc["cov6"]))
    with torch.no_grad():
        eager, radii = GaussianRasterizer(settings)(**args)  # (the shape is known from here on: a capture is sized from its hint)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            GaussianRasterizer(settings)(**args)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            img, rad = GaussianRasterizer(settings)(**args)
        assert img.shape == eager.shape == (3, 40, 48) and rad.shape == radii.shape == (400,)
        img.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(img, eager) and torch.equal(rad, radii)
        # hooks
        seen = []
        rast = GaussianRasterizer(settings)
        rast.register_forward_hook(lambda mod, a, out: seen.append("module"))
        rast(**args)
        h = torch.nn.modules.module.register_module_forward_hook(lambda mod, a, out: seen.append("global"))
        try:
            GaussianRasterizer(settings)(**args)
        finally:
            h.remove()
        assert seen == ["module", "global"]
        out_plain, _ = GaussianRasterizer(settings)(**args)  # (and without any hook: the short path, the same image)
        assert torch.equal(out_plain, eager)
