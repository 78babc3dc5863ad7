"""F3 oracle self-consistency: the analytic backward of the fp64 oracle is the exact gradient of its forward
(central finite differences), and the fp32 instantiation agrees with the fp64 one."""
import numpy as np
import pytest

from oracle import OracleRasterizer
from tests.util import look_at_c2w, make_camera, random_small_scene, rel_l2

H = W = 24


def _setup(seed, n, sh_coeffs, use_scale_rot=False, bg=(0.3, 0.1, 0.6), with_extra=True):
    sc = random_small_scene(seed, n, sh_coeffs=sh_coeffs)
    cam = make_camera(look_at_c2w((0.4, -0.3, -0.5)), near=0.5)
    rng = np.random.default_rng(seed + 100)
    wc = rng.uniform(0, 1, (3, H, W))
    we = rng.uniform(0, 1, (H, W)) * 0.1
    deg = {0: 0, 1: 0, 4: 1, 9: 2, 16: 3, 25: 4}[sh_coeffs]
    extra = sc["means"][:, 2] * 0.3 + 0.1 if with_extra else None

    def run(o, p):
        kw = dict(height=H, width=W, bg=bg, sh_degree=deg, means3D=p["means"], opacities=p["opac"], extra=p.get("extra"), **cam)
        if use_scale_rot:
            kw.update(scales=p["scales"], rotations=p["rots"])
        else:
            kw.update(cov3D_precomp=p["cov6"])
        if sh_coeffs > 0:
            kw["shs"] = p["colors"]
        else:
            kw["colors_precomp"] = p["colors"]
        return o.forward(**kw)

    def loss(res):
        l = float((res.color * wc).sum())
        if res.extra is not None:
            l += float((res.extra * we).sum())
        return l

    params = dict(sc)
    if with_extra:
        params["extra"] = extra
    return params, run, loss, wc, we


def _fd(params, run, loss, name, idxs, eps):
    o = OracleRasterizer(np.float64)
    out = []
    for idx in idxs:
        p = {k: (v.copy() if v is not None else None) for k, v in params.items()}
        p[name][idx] += eps
        lp = loss(run(o, p))
        p[name][idx] -= 2 * eps
        lm = loss(run(o, p))
        out.append((lp - lm) / (2 * eps))
    return np.array(out)


@pytest.mark.parametrize("sh_coeffs", [25, 9, 0])
def test_fp64_backward_is_gradient_of_forward(sh_coeffs):
    params, run, loss, wc, we = _setup(7, 12, sh_coeffs)
    o = OracleRasterizer(np.float64)
    res = run(o, params)
    assert res.n_visible == 12 and res.pairs_blended > 500
    g = o.backward(wc, we)
    rng = np.random.default_rng(0)
    checks = {
        "means": ("means3D", [(i, j) for i in range(12) for j in range(3)], 1e-6),
        "cov6": ("cov3D_precomp", [(int(i), int(j)) for i, j in zip(rng.integers(0, 12, 20), rng.integers(0, 6, 20))], 1e-6),
        "opac": ("opacities", [(i,) for i in range(12)], 1e-6),
        "colors": ("colors", [tuple(int(t) for t in rng.integers(0, s)) for _ in range(30)
                              for s in [params["colors"].shape]], 1e-6),
        "extra": ("extra", [(i,) for i in range(12)], 1e-6),
    }
    for pname, (gname, idxs, eps) in checks.items():
        fd = _fd(params, run, loss, pname, idxs, eps)
        an = np.array([g[gname][idx] for idx in idxs])
        err = np.abs(fd - an).max() / max(np.abs(fd).max(), 1e-12)
        assert err < 2e-6, (pname, err, fd[:4], an[:4])


def test_fp64_backward_scale_rotation():
    params, run, loss, wc, we = _setup(11, 10, 0, use_scale_rot=True)
    o = OracleRasterizer(np.float64)
    run(o, params)
    g = o.backward(wc, we)
    for pname, gname, shape in (("scales", "scales", (10, 3)), ("rots", "rotations", (10, 4))):
        idxs = [(i, j) for i in range(shape[0]) for j in range(shape[1])]
        fd = _fd(params, run, loss, pname, idxs, 1e-7)
        an = np.array([g[gname][idx] for idx in idxs])
        err = np.abs(fd - an).max() / max(np.abs(fd).max(), 1e-12)
        assert err < 5e-6, (pname, err)


def test_fp32_oracle_tracks_fp64_oracle():
    params, run, loss, wc, we = _setup(21, 64, 25)
    o64, o32 = OracleRasterizer(np.float64), OracleRasterizer(np.float32)
    r64, r32 = run(o64, params), run(o32, params)
    assert rel_l2(r32.color, r64.color) < 2e-6
    assert rel_l2(r32.extra, r64.extra) < 2e-6
    np.testing.assert_array_equal(r32.radii, r64.radii)
    g64, g32 = o64.backward(wc, we), o32.backward(wc, we)
    for k in ("means3D", "cov3D_precomp", "opacities", "colors", "extra", "means2D"):
        assert rel_l2(g32[k], g64[k]) < 2e-4, k


def test_threads_do_not_change_forward_and_barely_change_backward():
    params, run, loss, wc, we = _setup(5, 200, 25)
    o1, o4 = OracleRasterizer(np.float32, threads=1), OracleRasterizer(np.float32, threads=4)
    r1, r4 = run(o1, params), run(o4, params)
    np.testing.assert_array_equal(r1.color, r4.color)
    g1, g4 = o1.backward(wc, we), o4.backward(wc, we)
    for k in ("means3D", "cov3D_precomp", "opacities", "colors"):
        assert rel_l2(g4[k], g1[k]) < 1e-5, k


def test_means2d_gradient_is_screen_space_times_half_extent():
    """dL/dmeans2D is the pixel-space gradient (x W/2, x H/2 relative to NDC), SURVEY.md §8b."""
    params, run, loss, wc, we = _setup(9, 6, 0, with_extra=False)
    o = OracleRasterizer(np.float64)
    run(o, params)
    g = o.backward(wc, None)
    assert np.all(g["means2D"][:, 2] == 0) and np.abs(g["means2D"][:, :2]).max() > 0
