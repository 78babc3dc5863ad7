"""SURVEY.md 8f-4 `rotate_sh`.  basis="rasterizer": in the basis the rasterizer evaluates, rotating the coefficients is the same as rotating the
argument; the bands do not mix, rotations compose, and a rotated scene rendered with rotated coefficients gives the image of
the unrotated one (oracle rasterizer).  basis="e3nn" (the default: the matrices the reference applies,
src/misc/sh_rotation.py:24-34): rebuilt from e3nn's conventions, checked against what can be checked without e3nn."""
import numpy as np
import torch

from oracle import OracleRasterizer
from functools import partial

from pf3plat_amd import sh_rotation
from pf3plat_amd.sh_rotation import band_rotations, e3nn_band_rotations, sh_basis

rotate_sh = partial(sh_rotation.rotate_sh, basis="rasterizer")
from tests.util import make_camera, random_small_scene


def _rot(seed):
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=torch.Generator().manual_seed(seed), dtype=torch.float64))
    return q * torch.sign(torch.det(q))


def test_basis_is_the_one_the_oracle_evaluates():
    """A single Gaussian coloured by one coefficient at a time: oracle colour - 0.5 == sh_basis(direction)."""
    cam = make_camera()
    mean = np.array([[0.3, -0.2, 4.0]])
    d = mean[0] / np.linalg.norm(mean[0])
    want = sh_basis(torch.tensor(d)).numpy()
    for k in range(25):
        sh = np.zeros((1, 25, 3))
        sh[0, k, :] = 1.0
        o = OracleRasterizer(np.float64)
        o.forward(height=8, width=8, bg=(0, 0, 0), sh_degree=4, means3D=mean, opacities=np.array([0.5]),
                  cov3D_precomp=np.array([[0.01, 0, 0, 0.01, 0, 0.01]]), shs=sh, **cam)
        rgb = o.geometry()["rgb"][0]
        assert abs(max(want[k] + 0.5, 0.0) - rgb[0]) < 1e-9, k


def test_rotating_coefficients_equals_rotating_the_argument():
    g = torch.Generator().manual_seed(0)
    c = torch.randn((5, 3, 25), generator=g, dtype=torch.float64)
    r = torch.stack([_rot(s) for s in range(5)])[:, None]  # (5, 1, 3, 3) broadcast over the channel axis
    rc = rotate_sh(c, r)
    d = torch.nn.functional.normalize(torch.randn((5, 40, 3), generator=g, dtype=torch.float64), dim=-1)
    world = d @ r[:, 0].transpose(-1, -2)  # R d
    lhs = torch.einsum("bpn,bcn->bpc", sh_basis(world), rc)
    rhs = torch.einsum("bpn,bcn->bpc", sh_basis(d), c)
    np.testing.assert_allclose(lhs.numpy(), rhs.numpy(), atol=1e-10)
    # band 0 untouched; identity rotation is the identity; rotations compose; band matrices are orthogonal
    np.testing.assert_allclose(rc[..., 0].numpy(), c[..., 0].numpy(), atol=1e-12)
    np.testing.assert_allclose(rotate_sh(c, torch.eye(3, dtype=torch.float64)).numpy(), c.numpy(), atol=1e-10)
    a, b = _rot(7), _rot(8)
    np.testing.assert_allclose(rotate_sh(rotate_sh(c, b), a).numpy(), rotate_sh(c, a @ b).numpy(), atol=1e-9)
    for m in band_rotations(a, 4):
        np.testing.assert_allclose((m @ m.T).numpy(), np.eye(m.shape[0]), atol=1e-9)
    # float32 in, float32 out; lower degrees
    assert rotate_sh(c[..., :9].float(), r.float()).dtype == torch.float32


def test_rotated_scene_with_rotated_coefficients_renders_the_same_image():
    sc = random_small_scene(3, 60, sh_coeffs=25, dtype=np.float64)
    cam = make_camera(dtype=np.float64)
    r = _rot(21).numpy()
    t = np.array([0.4, -0.3, 0.2])

    def render(means, cov6, shs, c2w):
        o = OracleRasterizer(np.float64)
        return o.forward(height=24, width=32, bg=(0.1, 0.2, 0.3), sh_degree=4, means3D=means, opacities=sc["opac"],
                         cov3D_precomp=cov6, shs=shs, **make_camera(c2w, dtype=np.float64)).color

    base = render(sc["means"], sc["cov6"], sc["colors"], np.eye(4))
    cov = np.zeros((60, 3, 3))
    for k, (i, j) in enumerate(((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))):
        cov[:, i, j] = cov[:, j, i] = sc["cov6"][:, k]
    cov_r = r @ cov @ r.T
    cov6_r = np.stack([cov_r[:, 0, 0], cov_r[:, 0, 1], cov_r[:, 0, 2], cov_r[:, 1, 1], cov_r[:, 1, 2], cov_r[:, 2, 2]], -1)
    shs_r = rotate_sh(torch.tensor(sc["colors"]).permute(0, 2, 1), torch.tensor(r)).permute(0, 2, 1).numpy()  # (N, M, 3) <-> (N, 3, M)
    c2w = np.eye(4)
    c2w[:3, :3], c2w[:3, 3] = r, t
    moved = render(sc["means"] @ r.T + t, cov6_r, shs_r, c2w)
    np.testing.assert_allclose(moved, base, atol=2e-6)


def _e3nn_harmonics(d):
    """e3nn's real harmonics for l = 1..3 as its generated code states them (o3/_spherical_harmonics.py, 'component'
    normalisation; the per-band constant does not matter for the representation matrices): polar axis y."""
    x, y, z = d.unbind(-1)
    s1 = [x, y, z]
    s20, s21, s23, s24 = 15 ** 0.5 * x * z, 15 ** 0.5 * x * y, 15 ** 0.5 * y * z, 0.5 * 15 ** 0.5 * (z * z - x * x)
    s22 = 5 ** 0.5 * (y * y - 0.5 * (x * x + z * z))
    s2 = [s20, s21, s22, s23, s24]
    s3 = [(1 / 6) * 42 ** 0.5 * (s20 * z + s24 * x), 7 ** 0.5 * s20 * y, (1 / 8) * 168 ** 0.5 * (4 * y * y - (x * x + z * z)) * x,
          0.5 * 7 ** 0.5 * y * (2 * y * y - 3 * (x * x + z * z)), (1 / 8) * 168 ** 0.5 * z * (4 * y * y - (x * x + z * z)),
          7 ** 0.5 * s24 * y, (1 / 6) * 42 ** 0.5 * (s24 * z - s20 * x)]
    return [torch.stack(b, -1) for b in (s1, s2, s3)]


def test_e3nn_convention_band_matrices():
    """What the reference multiplies the coefficients with: D_l(R) of e3nn's real basis.  Without e3nn: (i) its l = 1 irrep is the
    vector representation in (x, y, z) order, so band 1 must be R itself; (ii) D_l(R) Y(d) = Y(R d) for e3nn's own polynomial
    forms, l <= 3; (iii) group law and orthogonality for all bands; (iv) it is NOT the rasterizer-basis rotation for l >= 1."""
    g = torch.Generator().manual_seed(1)
    d = torch.nn.functional.normalize(torch.randn((50, 3), generator=g, dtype=torch.float64), dim=-1)
    a, b = _rot(11), _rot(12)
    ma, mb, mab = e3nn_band_rotations(a, 4), e3nn_band_rotations(b, 4), e3nn_band_rotations(a @ b, 4)
    np.testing.assert_allclose(ma[0].numpy(), np.ones((1, 1)), atol=1e-12)
    np.testing.assert_allclose(ma[1].numpy(), a.numpy(), atol=1e-10)
    for l, (y, yr) in enumerate(zip(_e3nn_harmonics(d), _e3nn_harmonics(d @ a.T)), start=1):
        np.testing.assert_allclose((y @ ma[l].T).numpy(), yr.numpy(), atol=1e-9, err_msg=f"band {l}")
    for l in range(5):
        np.testing.assert_allclose((ma[l] @ mb[l]).numpy(), mab[l].numpy(), atol=1e-9)
        np.testing.assert_allclose((ma[l] @ ma[l].T).numpy(), np.eye(2 * l + 1), atol=1e-9)
    c = torch.randn((4, 3, 25), generator=g, dtype=torch.float64)
    ref, ras = sh_rotation.rotate_sh(c, a), rotate_sh(c, a)
    np.testing.assert_allclose(ref[..., 0].numpy(), c[..., 0].numpy(), atol=1e-12)
    assert not np.allclose(ref[..., 1:4].numpy(), ras[..., 1:4].numpy(), atol=1e-3)
    np.testing.assert_allclose(ref[..., 1:4].numpy(), (c[..., 1:4] @ a.T).numpy(), atol=1e-10)  # band 1: plain R c
    # a matrix that is not a rotation: the reference substitutes the identity (sh_rotation.py:21-22)
    np.testing.assert_allclose(sh_rotation.rotate_sh(c, 2.0 * a).numpy(), c.numpy())


def test_default_basis_warns_once_on_harmonics_read_from_a_ply(tmp_path):
    """`rotate_sh` without a basis on harmonics that came from `gaussians_from_ply` (a view of them included) warns - once - that the
    reference's e3nn convention is about to be applied to coefficients in the rasterizer's basis; an explicit basis is silent, and
    harmonics from anywhere else never warn."""
    import warnings

    import numpy as np

    from pf3plat_amd import ply_export, sh_rotation

    g = 5
    rng = np.random.default_rng(0)
    d = dict(xyz=rng.normal(size=(g, 3)).astype(np.float32), f_dc=rng.normal(size=(g, 3)).astype(np.float32),
             f_rest=rng.normal(size=(g, 3 * 24)).astype(np.float32), opacity=rng.uniform(size=g).astype(np.float32),
             scale=rng.normal(size=(g, 3)).astype(np.float32), rot=np.tile(np.array([1, 0, 0, 0], np.float32), (g, 1)))
    gs = ply_export.gaussians_from_ply(d)
    rot = torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    sh_rotation._warned_external = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        sh_rotation.rotate_sh(torch.randn(4, 25), rot)  # not from a ply: silent
        sh_rotation.rotate_sh(gs.harmonics[0], rot, basis="rasterizer")  # explicit: silent
        sh_rotation.rotate_sh(gs.harmonics[0], rot, basis="e3nn")
        assert not w
        out = sh_rotation.rotate_sh(gs.harmonics[0], rot)  # a view of the registered tensor, default basis
        assert len(w) == 1 and "basis='rasterizer'" in str(w[0].message)
        sh_rotation.rotate_sh(gs.harmonics, rot)  # once per process
        assert len(w) == 1
    assert torch.equal(out, sh_rotation.rotate_sh(gs.harmonics[0], rot, basis="e3nn"))  # the default is still the reference's convention
