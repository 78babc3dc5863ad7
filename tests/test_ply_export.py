"""`export_ply` against the vertex table the REFERENCE's export_ply produces for the same seeded Gaussians
(tests/golden/ply_fixture.npz, recorded by tests/golden/make_ply_fixture.py with plyfile replaced by a recorder), and the
on-disk format through a write / read round trip."""
import os

import numpy as np
import torch

from pf3plat_amd import ply_export

FIX = np.load(os.path.join(os.path.dirname(__file__), "golden", "ply_fixture.npz"))
t = lambda k: torch.tensor(FIX[k])


def _args():
    return t("ext"), t("means"), t("scales"), t("rotations"), t("harmonics"), t("opacities")


def test_vertex_table_matches_the_reference():
    assert tuple(FIX["names"]) == ply_export.PROPERTIES and str(FIX["element"]) == "vertex"
    got, want = ply_export.vertex_table(*_args()), FIX["table"]
    np.testing.assert_allclose(got[:, :13], want[:, :13], rtol=2e-5, atol=2e-6)  # xyz, normals, DC, opacity, log scales
    # quaternions (w, x, y, z): q and -q are the same rotation
    sign = np.sign((got[:, 13:] * want[:, 13:]).sum(-1, keepdims=True))
    np.testing.assert_allclose(got[:, 13:] * sign, want[:, 13:], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(np.linalg.norm(got[:, 13:], axis=-1), 1.0, atol=1e-6)


def test_round_trip_through_the_file(tmp_path):
    path = tmp_path / "sub" / "scene.ply"
    ply_export.export_ply(*_args(), path)
    raw = path.read_bytes()
    assert raw.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 50\nproperty float x\n")
    assert len(raw) == raw.index(b"end_header\n") + len(b"end_header\n") + 50 * 17 * 4
    back = ply_export.read_ply(path)
    table = ply_export.vertex_table(*_args())
    np.testing.assert_array_equal(back["xyz"], table[:, 0:3])
    np.testing.assert_array_equal(back["f_dc"], table[:, 6:9])
    np.testing.assert_array_equal(back["opacity"], table[:, 9])
    np.testing.assert_array_equal(back["scale"], table[:, 10:13])
    np.testing.assert_array_equal(back["rot"], table[:, 13:17])
    assert back["f_rest"].shape == (50, 0)


def test_quaternion_matrix_round_trip():
    q = np.random.default_rng(0).normal(size=(200, 4))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    back = ply_export.matrix_to_quaternion_xyzw(ply_export.quaternion_xyzw_to_matrix(q))
    sign = np.sign((back * q).sum(-1, keepdims=True))
    np.testing.assert_allclose(back * sign, q, atol=1e-12)


def _ply_scene(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    means = torch.stack((torch.rand(n, generator=g) * 6 - 3, torch.rand(n, generator=g) * 6 - 3, torch.rand(n, generator=g) * 6 + 3), -1)
    scales = 0.03 + 0.12 * torch.rand((n, 3), generator=g)
    quats = torch.nn.functional.normalize(torch.randn((n, 4), generator=g), dim=-1)
    harmonics = torch.rand((n, 3, 1), generator=g) * 2 - 0.5  # DC only: what the file keeps
    opac = 0.2 + 0.75 * torch.rand(n, generator=g)
    return means, scales, quats, harmonics, opac


def _render_pair(device, tmp_path):
    """The same scene rendered directly and after export_ply -> read_ply -> gaussians_from_ply, looked at through the camera the
    export's similarity transform (median shift, rescale, viewer rotation) maps the original camera to."""
    import pf3plat_amd
    from pf3plat_amd.ply_export import export_ply, gaussians_from_ply, viewer_rotation
    from pf3plat_amd.types import Gaussians

    n, hw = 4000, (64, 64)
    means, scales, quats, harmonics, opac = _ply_scene(n)
    ext = torch.eye(4)
    ang = 0.2
    ext[0, 0], ext[0, 2], ext[2, 0], ext[2, 2] = np.cos(ang), np.sin(ang), -np.sin(ang), np.cos(ang)
    ext[:3, 3] = torch.tensor([0.3, -0.2, -0.5])
    intr = torch.tensor([[0.9, 0, 0.5], [0, 0.9, 0.5], [0, 0, 1.0]])
    path = tmp_path / "scene.ply"
    export_ply(ext, means, scales, quats, harmonics, opac, path)
    back = gaussians_from_ply(path, device=device)
    assert back.covariances is None and back.means.shape == (1, n, 3) and back.harmonics.shape == (1, n, 3, 1)
    # the similarity the export applied: x' = R (x - median) / unit
    shifted = means - means.median(dim=0).values
    unit = float(shifted.abs().quantile(0.95, dim=0).max())
    rot = torch.tensor(viewer_rotation(ext), dtype=torch.float32)
    ext2 = torch.eye(4)
    ext2[:3, :3] = rot @ ext[:3, :3]
    ext2[:3, 3] = rot @ ((ext[:3, 3] - means.median(dim=0).values) / unit)
    near, far = torch.tensor([[0.5]]), torch.tensor([[50.0]])
    dec = pf3plat_amd.DecoderSplattingCUDA().to(device)
    to = lambda x: x.to(device)
    direct = Gaussians(to(means)[None], None, to(harmonics)[None], to(opac)[None], scales=to(scales)[None], rotations=to(quats)[None])
    a = dec.forward(direct, to(ext)[None, None], to(intr)[None, None], to(near), to(far), hw, depth_mode="depth")
    b = dec.forward(back, to(ext2)[None, None], to(intr)[None, None], to(near / unit), to(far / unit), hw, depth_mode="depth")
    return a, b, unit


def test_exported_scene_read_back_renders_the_same_image(oracle_backend, tmp_path):
    a, b, unit = _render_pair("cpu", tmp_path)
    assert a.color.abs().max() > 0.2
    np.testing.assert_allclose(b.color.numpy(), a.color.numpy(), atol=3e-5)
    np.testing.assert_allclose(b.depth.numpy() * unit, a.depth.numpy(), rtol=1e-4, atol=1e-4)  # depth comes back in file units


import pytest  # noqa: E402


@pytest.mark.gpu
def test_hip_renders_an_exported_ply_like_the_scene_it_came_from(tmp_path):
    """SURVEY 8f-4 on the device: export_ply -> read_ply -> gaussians_from_ply -> DecoderSplattingCUDA on the HIP library (the
    scale + quaternion form, gsr_forward_scale_rot) gives the image and the depth of the original Gaussians seen through the
    correspondingly transformed camera - and both agree with the oracle-driven CPU run of the same code."""
    from tests.oracle_backend import OracleBackend
    from tests.util import install_backend

    a, b, unit = _render_pair("cuda:0", tmp_path)
    assert a.color.abs().max() > 0.2
    np.testing.assert_allclose(b.color.cpu().numpy(), a.color.cpu().numpy(), atol=3e-5)
    np.testing.assert_allclose(b.depth.cpu().numpy() * unit, a.depth.cpu().numpy(), rtol=1e-4, atol=1e-4)
    old = install_backend(OracleBackend(threads=8))
    try:
        oa, ob, _ = _render_pair("cpu", tmp_path)
    finally:
        install_backend(old)
    np.testing.assert_allclose(a.color.cpu().numpy(), oa.color.numpy(), atol=2e-5)
    np.testing.assert_allclose(b.color.cpu().numpy(), ob.color.numpy(), atol=2e-5)
