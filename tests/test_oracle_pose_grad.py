"""the oracle's camera gradient (viewmatrix, projmatrix, campos as the operator's independent inputs) is the
exact gradient of its fp64 forward - central finite differences over every camera entry the forward reads."""
import numpy as np
import pytest

from oracle import OracleRasterizer
from tests.test_oracle_grad import H, W, _setup


def _camera_fd(params, run_with, loss, name, n, eps):
    o = OracleRasterizer(np.float64)
    out = np.zeros(n)
    for k in range(n):
        lp = loss(run_with(o, name, k, +eps))
        lm = loss(run_with(o, name, k, -eps))
        out[k] = (lp - lm) / (2 * eps)
    return out


@pytest.mark.parametrize("sh_coeffs", [25, 9, 0])
def test_fp64_camera_gradient_is_gradient_of_forward(sh_coeffs):
    params, run, loss, wc, we = _setup(7, 12, sh_coeffs)
    # `run` closes over the camera dict: rebuild the call with a perturbed copy
    from tests.util import look_at_c2w, make_camera
    cam = make_camera(look_at_c2w((0.4, -0.3, -0.5)), near=0.5)
    deg = {0: 0, 9: 2, 25: 4}[sh_coeffs]

    def run_with(o, name, k, delta):
        c = {key: (np.array(val, dtype=np.float64).copy() if isinstance(val, np.ndarray) else val) for key, val in cam.items()}
        if name is not None:
            flat = c[name].reshape(-1)
            flat[k] += delta
        kw = dict(height=H, width=W, bg=(0.3, 0.1, 0.6), sh_degree=deg, means3D=params["means"], opacities=params["opac"],
                  extra=params["extra"], cov3D_precomp=params["cov6"], **c)
        kw["shs" if sh_coeffs > 0 else "colors_precomp"] = params["colors"]
        return o.forward(**kw)

    o = OracleRasterizer(np.float64)
    res = run_with(o, None, 0, 0.0)
    assert res.n_visible == 12
    g = o.backward(wc, we)["camera"]
    for name, lo, n in (("viewmatrix", 0, 16), ("projmatrix", 16, 16), ("campos", 32, 3)):
        fd = _camera_fd(params, run_with, loss, name, n, 1e-6)
        an = g[lo:lo + n]
        scale = max(np.abs(fd).max(), 1e-12)
        if name == "campos" and sh_coeffs == 0:
            assert np.all(an == 0) and np.abs(fd).max() < 1e-9
            continue
        assert np.abs(fd - an).max() / scale < 2e-6, (name, fd, an)
        # entries the forward never reads: the 4th row of the view matrix, the depth row of the projection
        unread = [3, 7, 11, 15] if name == "viewmatrix" else [2, 6, 10, 14] if name == "projmatrix" else []
        assert all(an[k] == 0 for k in unread)
    assert np.abs(g[:16]).max() > 1e-3 and np.abs(g[16:32]).max() > 1e-3


def test_fp32_camera_gradient_agrees_with_fp64():
    params, run, loss, wc, we = _setup(11, 40, 25)
    g = {}
    for dt in (np.float32, np.float64):
        o = OracleRasterizer(dt)
        run(o, params)
        g[dt] = o.backward(wc.astype(dt), we.astype(dt))["camera"].astype(np.float64)
    for lo, hi in ((0, 16), (16, 32), (32, 35)):
        a, b = g[np.float32][lo:hi], g[np.float64][lo:hi]
        assert np.abs(a - b).max() <= 2e-4 * np.abs(b).max(), (lo, a, b)
