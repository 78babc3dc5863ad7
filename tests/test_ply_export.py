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
