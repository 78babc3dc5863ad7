"""F2 analytic known-answer tests for the CPU oracle: closed-form answers that do not
need the (unavailable) reference rasterizer.  These are the pins of an otherwise "parity unpinned" oracle."""
import math

import numpy as np
import pytest

from oracle import OracleRasterizer
from tests.util import make_camera

SH_C0 = 0.28209479177387814


def _single(o, z0=4.0, sigma=0.2, opacity=0.6, dc=(1.0, 0.2, -0.5), W=32, H=32, bg=(0.1, 0.2, 0.3), mean_xy=(0.0, 0.0), **kw):
    cam = make_camera()
    shs = np.zeros((1, 1, 3))
    shs[0, 0] = dc
    cov6 = np.array([[sigma ** 2, 0, 0, sigma ** 2, 0, sigma ** 2]])
    return o.forward(height=H, width=W, bg=bg, sh_degree=0, means3D=np.array([[mean_xy[0], mean_xy[1], z0]]),
                     opacities=np.array([opacity]), cov3D_precomp=cov6, shs=shs, **cam, **kw), cam


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_single_isotropic_gaussian_closed_form(dtype):
    W = H = 32
    o = OracleRasterizer(dtype)
    z0, sigma, opacity, dc, bg = 4.0, 0.2, 0.6, (1.0, 0.2, -0.5), (0.1, 0.2, 0.3)
    res, cam = _single(o, z0, sigma, opacity, dc, W, H, bg)
    fx = W / (2 * cam["tanfovx"])
    a = (fx * sigma / z0) ** 2 + 0.3  # EWA + 0.3 px^2 dilation
    lam = a + math.sqrt(0.1)  # mid + sqrt(max(0.1, mid^2 - det)), isotropic => mid^2 - det = 0
    assert res.radii[0] == math.ceil(3 * math.sqrt(lam))
    geo = o.geometry()
    np.testing.assert_allclose(geo["xy"][0], [(W - 1) / 2, (H - 1) / 2], atol=1e-5)
    np.testing.assert_allclose(geo["conic_opacity"][0], [1 / a, 0, 1 / a, opacity], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(geo["depth"][0], z0, rtol=1e-6)
    rgb = np.maximum(SH_C0 * np.array(dc) + 0.5, 0)
    np.testing.assert_allclose(geo["rgb"][0], rgb, rtol=1e-6)
    ys, xs = np.mgrid[0:H, 0:W]
    r2 = (xs - (W - 1) / 2) ** 2 + (ys - (H - 1) / 2) ** 2
    alpha = np.minimum(0.99, opacity * np.exp(-0.5 * r2 / a))
    alpha = np.where(alpha < 1 / 255, 0, alpha)
    # only pixels in 16x16 tiles touched by the ceil(3 sigma) rect see the splat
    r = res.radii[0]
    cx = (W - 1) / 2
    t0, t1 = int((cx - r) / 16), int((cx + r + 15) / 16)
    mask = np.zeros((H, W), bool)
    mask[t0 * 16:t1 * 16, t0 * 16:t1 * 16] = True
    alpha = np.where(mask, alpha, 0)
    expect = rgb[:, None, None] * alpha + (1 - alpha) * np.array(bg)[:, None, None]
    np.testing.assert_allclose(res.color, expect, rtol=2e-5 if dtype == np.float32 else 1e-9, atol=1e-6)


def test_near_plane_cull_is_at_0p2():
    o = OracleRasterizer(np.float64)
    res, _ = _single(o, z0=0.2)
    assert res.radii[0] == 0 and res.n_visible == 0
    np.testing.assert_allclose(res.color, np.array([0.1, 0.2, 0.3])[:, None, None] * np.ones((3, 32, 32)))
    res, _ = _single(o, z0=0.2001, sigma=0.01)
    assert res.radii[0] > 0


def test_offscreen_gaussian_is_culled_by_empty_rect():
    o = OracleRasterizer(np.float64)
    res, _ = _single(o, mean_xy=(50.0, 0.0))
    assert res.radii[0] == 0 and res.r16 == 0


def test_alpha_cap_and_threshold():
    o = OracleRasterizer(np.float64)
    W = H = 32
    # opacity 1.0: centre pixel alpha = min(0.99, exp(-q/2)); choose the mean so that a pixel centre coincides
    cam = make_camera()
    # pixel (16,16) <-> ndc ((2*16+1)/32 - 1) => x = ndc * tanfov * z
    z0 = 4.0
    x0 = ((2 * 16 + 1) / W - 1) * cam["tanfovx"] * z0
    res, _ = _single(o, z0=z0, opacity=1.0, dc=(1.0, 1.0, 1.0), bg=(0, 0, 0), mean_xy=(x0, x0))
    rgb = SH_C0 + 0.5
    np.testing.assert_allclose(res.color[:, 16, 16], 0.99 * rgb, rtol=1e-6)
    # opacity below 1/255 contributes nothing at all
    res, _ = _single(o, z0=z0, opacity=1 / 255 - 1e-6, bg=(0, 0, 0))
    assert np.all(res.color == 0) and res.pairs_blended == 0
    res, _ = _single(o, z0=z0, opacity=1 / 255 + 1e-4, bg=(0, 0, 0), mean_xy=(x0, x0))
    assert res.pairs_blended >= 1


def test_depth_order_and_index_tie_break():
    o = OracleRasterizer(np.float64)
    cam = make_camera()
    W = H = 16
    cov6 = np.tile(np.array([[0.5, 0, 0, 0.5, 0, 0.5]]), (2, 1))
    col = np.array([[1.0, 0, 0], [0, 0, 1.0]])
    kw = dict(height=H, width=W, bg=(0, 0, 0), sh_degree=0, opacities=np.array([0.8, 0.8]), cov3D_precomp=cov6,
              colors_precomp=col, **cam)
    front_red = o.forward(means3D=np.array([[0, 0, 3.0], [0, 0, 5.0]]), **kw).color[:, 8, 8]
    front_blue = o.forward(means3D=np.array([[0, 0, 5.0], [0, 0, 3.0]]), **kw).color[:, 8, 8]
    assert front_red[0] > front_red[2] and front_blue[2] > front_blue[0]
    # identical depth: lower index is composited first
    tie = o.forward(means3D=np.array([[0, 0, 4.0], [0, 0, 4.0]]), **kw).color[:, 8, 8]
    assert tie[0] > tie[2]
    np.testing.assert_allclose(o.binning()["point_list"][:2], [0, 1])


def test_transmittance_early_out_and_state():
    o = OracleRasterizer(np.float64)
    cam = make_camera()
    n, W, H = 40, 16, 16
    means = np.stack([np.zeros(n), np.zeros(n), np.linspace(2, 6, n)], -1)
    cov6 = np.tile(np.array([[4.0, 0, 0, 4.0, 0, 4.0]]), (n, 1))
    res = o.forward(height=H, width=W, bg=(0, 0, 0), sh_degree=0, means3D=means, opacities=np.full(n, 0.9),
                    cov3D_precomp=cov6, colors_precomp=np.ones((n, 3)), **cam)
    st = o.image_state()
    # alpha = 0.9 each (huge footprint): T after k splats = 0.1^k; stops when T(1-a) < 1e-4 => 4 contributors
    assert st["n_contrib"][8, 8] == 4
    # (alpha is a little below 0.9 half a pixel off-centre, so T ends slightly above 1e-4)
    assert 1e-4 <= st["final_T"][8, 8] < 1.2e-4
    np.testing.assert_allclose(res.color[:, 8, 8], 1 - st["final_T"][8, 8], rtol=1e-12)


def test_empty_input_renders_zeros_not_background():
    o = OracleRasterizer(np.float32)
    cam = make_camera()
    res = o.forward(height=8, width=8, bg=(0.5, 0.5, 0.5), sh_degree=0, means3D=np.zeros((0, 3)), opacities=np.zeros((0,)),
                    cov3D_precomp=np.zeros((0, 6)), colors_precomp=np.zeros((0, 3)), **cam)
    assert np.all(res.color == 0)


def test_sh_clamp_and_mask():
    o = OracleRasterizer(np.float64)
    res, _ = _single(o, dc=(-5.0, 0.0, 5.0))
    geo = o.geometry()
    assert list(geo["clamped"][0]) == [1, 0, 0]
    assert geo["rgb"][0, 0] == 0.0


@pytest.mark.parametrize("max_sh_eval", [3, 4])
def test_sh_bands_match_scipy_real_harmonics(max_sh_eval):
    """basis_k(dir) == (-1)^m * standard real SH Y_lm(dir) for every band the build evaluates (band 4 is
    gated by max_sh_eval: 3 = vanilla upstream, 4 = PF3plat's sh_degree=4 taken at face value)."""
    from scipy.special import sph_harm_y

    o = OracleRasterizer(np.float64, max_sh_eval=max_sh_eval)
    cam = make_camera()
    rng = np.random.default_rng(0)
    pos = np.array([[0.3, -0.2, 4.0]])
    d = pos[0] / np.linalg.norm(pos[0])
    theta, phi = math.acos(d[2]), math.atan2(d[1], d[0])
    for l in range(5):
        for m in range(-l, l + 1):
            k = l * l + l + m
            shs = np.zeros((1, 25, 3))
            shs[0, k, 0] = 1.0
            o.forward(height=16, width=16, bg=(0, 0, 0), sh_degree=4, means3D=pos, opacities=np.array([0.5]),
                      cov3D_precomp=np.array([[0.1, 0, 0, 0.1, 0, 0.1]]), shs=shs, **cam)
            got = o.geometry()["rgb"][0, 0] - 0.5
            y = sph_harm_y(l, abs(m), theta, phi)
            std = y.real if m == 0 else math.sqrt(2) * (-1) ** m * (y.real if m > 0 else y.imag)
            want = (-1) ** m * std if l <= max_sh_eval else 0.0
            assert abs(got - want) < 1e-12, (l, m, got, want)


def test_extra_channel_equals_second_pass_with_precomputed_colour():
    """The fused depth channel must equal what the reference's second raster pass produces
    (cuda_splatting.py:255-269: colour := f(z) on all three channels, bg 0, mean over channels)."""
    from tests.util import random_small_scene

    sc = random_small_scene(3, 40, sh_coeffs=9)
    cam = make_camera()
    o = OracleRasterizer(np.float64)
    extra = sc["means"][:, 2].copy()
    a = o.forward(height=32, width=32, bg=(0.2, 0.3, 0.4), sh_degree=2, means3D=sc["means"], opacities=sc["opac"],
                  cov3D_precomp=sc["cov6"], shs=sc["colors"], extra=extra, **cam)
    b = o.forward(height=32, width=32, bg=(0, 0, 0), sh_degree=0, means3D=sc["means"], opacities=sc["opac"],
                  cov3D_precomp=sc["cov6"], colors_precomp=np.repeat(extra[:, None], 3, 1), **cam)
    np.testing.assert_allclose(a.extra, b.color.mean(0), rtol=1e-12, atol=1e-14)


def test_scale_rotation_covariance_path():
    from tests.util import random_small_scene

    sc = random_small_scene(5, 20, sh_coeffs=0)
    cam = make_camera()
    o = OracleRasterizer(np.float64)
    a = o.forward(height=32, width=32, bg=(0, 0, 0), sh_degree=0, means3D=sc["means"], opacities=sc["opac"],
                  cov3D_precomp=sc["cov6"], colors_precomp=sc["colors"], **cam)
    b = o.forward(height=32, width=32, bg=(0, 0, 0), sh_degree=0, means3D=sc["means"], opacities=sc["opac"],
                  scales=sc["scales"], rotations=sc["rots"], colors_precomp=sc["colors"], **cam)
    np.testing.assert_allclose(a.color, b.color, rtol=1e-9, atol=1e-12)


def test_mean_projection_matches_reference_pixel_convention():
    """Appendix A.1 step 5 / Appendix C: the pixel centre equals project(p) * (W,H) - 0.5, with project() being
    the reference's pure-torch projector (src/geometry/projection.py:59-71: K @ (p_cam / z))."""
    rng = np.random.default_rng(1)
    W, H = 48, 32
    cam = make_camera(fx=0.8, fy=0.9)
    pts = np.stack([rng.uniform(-1, 1, 20), rng.uniform(-1, 1, 20), rng.uniform(2, 6, 20)], -1)
    o = OracleRasterizer(np.float64)
    o.forward(height=H, width=W, bg=(0, 0, 0), sh_degree=0, means3D=pts, opacities=np.full(20, 0.5),
              cov3D_precomp=np.tile(np.array([[0.01, 0, 0, 0.01, 0, 0.01]]), (20, 1)), colors_precomp=np.ones((20, 3)), **cam)
    xy = o.geometry()["xy"]
    k = np.array([[0.8, 0, 0.5], [0, 0.9, 0.5], [0, 0, 1]])
    uv = (k @ (pts / pts[:, 2:]).T).T[:, :2]
    np.testing.assert_allclose(xy, uv * np.array([W, H]) - 0.5, atol=1e-5)
