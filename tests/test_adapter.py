"""the encoder-side adapter and the scale / rotation input form of the raster path.

tests/golden/adapter_fixtures.npz holds what the REFERENCE's `GaussianAdapter.forward`, `build_covariance` and
`quaternion_to_matrix` return for seeded inputs (tests/golden/make_adapter_fixtures.py imports them on CPU in the build
container; `rotate_sh` replaced by a stand-in, so harmonics are recorded un-rotated).  Pinned here: the oracle's restatement
(oracle/adapter.py), the package's adapter (pf3plat_amd/adapter.py), and - through the oracle-driven wrappers on CPU and the
HIP path on the GPU - that rendering from scale + quaternion records equals rendering from the materialised covariances,
forward and backward."""
import os

import numpy as np
import pytest
import torch

import pf3plat_amd
from oracle import adapter as oracle_adapter
from pf3plat_amd import synthetic
from pf3plat_amd.adapter import GaussianAdapter, GaussianAdapterCfg
from pf3plat_amd.types import Gaussians
from tests.util import install_backend, rel_l2

FIX = np.load(os.path.join(os.path.dirname(__file__), "golden", "adapter_fixtures.npz"))
t = lambda k: torch.tensor(FIX[k])


def test_oracle_restatement_matches_reference_build_covariance():
    rot = oracle_adapter.rotation_from_quaternion_xyzw(t("A_quat"))
    np.testing.assert_allclose(rot.numpy(), FIX["A_rotation"], rtol=1e-6, atol=1e-6)
    records = torch.cat((t("A_scale"), t("A_quat")), -1)
    cov = oracle_adapter.covariance_from_scale_rotation(records)
    np.testing.assert_allclose(cov.numpy(), FIX["A_cov"], rtol=1e-5, atol=1e-7)
    # with per-group frames: C Sigma C^T, groups of N / F consecutive Gaussians
    frames = torch.linalg.qr(torch.randn(2, 4, 3, 3, generator=torch.Generator().manual_seed(0)))[0]
    cov_f = oracle_adapter.covariance_from_scale_rotation(records, frames)
    c = frames.repeat_interleave(10, dim=1)
    np.testing.assert_allclose(cov_f.numpy(), (c @ t("A_cov") @ c.transpose(-1, -2)).numpy(), rtol=1e-5, atol=1e-6)


def _adapted():
    lo, hi, deg = FIX["B_cfg"]
    ad = GaussianAdapter(GaussianAdapterCfg(float(lo), float(hi), int(deg)), rotate_sh=None)  # the fixture's harmonics are un-rotated
    hw = tuple(int(x) for x in FIX["B_hw"])
    raw = t("B_in_raw").requires_grad_(True)
    out = ad.forward(t("B_in_ext")[:, :, None], t("B_in_intr")[:, :, None], t("B_in_coords"), t("B_in_depths"), t("B_in_opac"), raw, hw)
    return ad, out, raw


def test_adapter_matches_reference_gaussian_adapter():
    ad, out, _ = _adapted()
    assert (ad.d_sh, ad.d_in) == (25, 82)
    np.testing.assert_allclose(ad.sh_mask.numpy(), FIX["B_sh_mask"], rtol=1e-7)
    for name, got in (("means", out.means), ("scales", out.scales), ("rotations", out.rotations), ("harmonics", out.harmonics),
                      ("opacities", out.opacities), ("covariances", out.covariances)):
        np.testing.assert_allclose(got.detach().numpy(), FIX["B_out_" + name], rtol=2e-5, atol=2e-7, err_msg=name)
    assert not out.frames.requires_grad
    np.testing.assert_allclose(out.frames.numpy(), FIX["B_in_ext"][:, :, :3, :3])
    g = out.for_decoder()
    assert g.covariances is None and g.means.shape == (1, 2 * 192, 3) and g.scales.shape == (1, 384, 3) and g.frames.shape == (1, 2, 3, 3)
    # by default the harmonics are rotated into world space, band by band (DC untouched)
    from pf3plat_amd.sh_rotation import rotate_sh
    lo, hi, deg = FIX["B_cfg"]
    hw = tuple(int(x) for x in FIX["B_hw"])
    rot = GaussianAdapter(GaussianAdapterCfg(float(lo), float(hi), int(deg))).forward(
        t("B_in_ext")[:, :, None], t("B_in_intr")[:, :, None], t("B_in_coords"), t("B_in_depths"), t("B_in_opac"), t("B_in_raw"), hw)
    want = rotate_sh(out.harmonics.detach(), out.frames[:, :, None, None])
    assert torch.allclose(rot.harmonics, want, atol=1e-6) and torch.equal(rot.harmonics[..., 0], out.harmonics[..., 0].detach())
    assert not torch.allclose(rot.harmonics[..., 1:], out.harmonics[..., 1:].detach(), atol=1e-3)


def test_adapter_accepts_the_encoders_call_shapes():
    """The reference encoder calls the adapter with extra batch dims behind the view axis (encoder_costvolume.py:529-540: cameras
    'b v () () () i j', depths 'b v r 1 1', opacities 'b v r srf spp') and keeps the first and last source view of the result
    (:556-573).  Fixture case C holds what the reference's adapter returns for such a call (3 views, 48 rays, 2 surfaces, 2
    samples; the reference leaves the sample axis of means / covariances at 1 - values are compared under broadcasting)."""
    lo, hi, deg = FIX["B_cfg"]
    ad = GaussianAdapter(GaussianAdapterCfg(float(lo), float(hi), int(deg)), rotate_sh=None)
    hw = tuple(int(x) for x in FIX["C_hw"])
    ext, intr = t("C_in_ext")[:, :, None, None, None], t("C_in_intr")[:, :, None, None, None]
    out = ad.forward(ext, intr, t("C_in_coords"), t("C_in_depths"), t("C_in_opac"), t("C_in_raw"), hw)
    full = FIX["C_in_opac"].shape
    assert out.frames.shape == (1, 3, 3, 3) and out.means.shape == (*full, 3) and out.harmonics.shape == (*full, 3, 25)
    for name, got in (("means", out.means), ("scales", out.scales), ("rotations", out.rotations), ("harmonics", out.harmonics),
                      ("opacities", out.opacities), ("covariances", out.covariances)):
        want = np.broadcast_to(FIX["C_out_" + name], got.shape)
        np.testing.assert_allclose(got.numpy(), want, rtol=2e-5, atol=1e-6 if name == "covariances" else 2e-7, err_msg=name)
    # what the encoder hands to the decoder: views (0, -1), everything behind the scene axis flattened view-major
    g = out.for_decoder(views=(0, -1))
    n = 2 * 48 * 2 * 2
    assert g.means.shape == (1, n, 3) and g.harmonics.shape == (1, n, 3, 25) and g.opacities.shape == (1, n) and g.frames.shape == (1, 2, 3, 3)
    np.testing.assert_allclose(g.means.numpy(), np.broadcast_to(FIX["C_out_means"], (*full, 3))[:, (0, -1)].reshape(1, n, 3), rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(g.frames.numpy(), FIX["C_in_ext"][:, (0, -1), :3, :3])
    assert out.for_decoder().means.shape == (1, 3 * 48 * 4, 3)
    with pytest.raises(ValueError, match="one camera per"):  # a camera per ray cannot become per-view frames
        ad.forward(ext.expand(1, 3, 48, 1, 1, 4, 4), intr, t("C_in_coords"), t("C_in_depths"), t("C_in_opac"), t("C_in_raw"), hw)


def _render(dec, g, sc, device="cpu"):
    mv = lambda x: x.to(device)
    return dec.forward(g, mv(sc.extrinsics), mv(sc.intrinsics), mv(sc.near), mv(sc.far), (24, 32), depth_mode="depth")


def _scene_and_weights():
    sc = synthetic.make_scene(3, 8, (24, 32), num_views=2)  # (only its cameras are used)
    w = torch.rand((1, 2, 3, 24, 32), generator=torch.Generator().manual_seed(5))
    wd = torch.rand((1, 2, 24, 32), generator=torch.Generator().manual_seed(6)) * 0.05
    return sc, w, wd


def test_decoder_renders_scale_rotation_records_like_materialised_covariances(oracle_backend):
    """Same image and same gradients w.r.t. the adapter's raw inputs, whether the decoder receives scale + quaternion records
    (covariance built inside the raster backend) or the (N, 3, 3) matrices the reference materialises."""
    sc, w, wd = _scene_and_weights()
    dec = pf3plat_amd.DecoderSplattingCUDA()
    res = []
    for fused in (True, False):
        _, out, raw = _adapted()
        g = out.for_decoder()
        if not fused:
            b, v, r = out.opacities.shape
            g = Gaussians(g.means, out.covariances.reshape(b, v * r, 3, 3), g.harmonics, g.opacities)
        o = _render(dec, g, sc)
        ((o.color * w).sum() + (o.depth * wd).sum()).backward()
        res.append((o.color.detach().numpy(), o.depth.detach().numpy(), raw.grad.numpy()))
    assert rel_l2(res[0][0], res[1][0]) < 1e-6 and rel_l2(res[0][1], res[1][1]) < 1e-6
    assert np.abs(res[1][2]).max() > 0 and rel_l2(res[0][2], res[1][2]) < 2e-5


@pytest.mark.gpu
def test_hip_scale_rotation_form_matches_the_oracle_forward_and_backward():
    """gsr_forward_scale_rot / gsr_backward_scale_rot (covariance built in registers on load, dL/dscale and dL/dquaternion
    returned directly) against the oracle rasterizer fed with the oracle's materialised covariances, through the same decoder."""
    from pf3plat_amd import rasterizer
    from tests.oracle_backend import OracleBackend

    sc, w, wd = _scene_and_weights()
    res = {}
    for dev in ("cuda:0", "cpu"):
        old = install_backend(OracleBackend(threads=8)) if dev == "cpu" else None
        try:
            ad, out, raw = _adapted()
            g = out.for_decoder()
            g = Gaussians(*[None if x is None else x.to(dev) for x in (g.means, None, g.harmonics, g.opacities, g.scales, g.rotations, g.frames)])
            dec = pf3plat_amd.DecoderSplattingCUDA().to(dev)
            o = _render(dec, g, sc, dev)
            ((o.color * w.to(dev)).sum() + (o.depth * wd.to(dev)).sum()).backward()
            res[dev] = (o.color.detach().cpu().numpy(), o.depth.detach().cpu().numpy(), raw.grad.numpy())
        finally:
            if dev == "cpu":
                install_backend(old)
    for k, name in enumerate(("colour", "depth", "d_raw")):
        assert rel_l2(res["cuda:0"][k], res["cpu"][k]) < 1e-4, name
    assert np.abs(res["cpu"][2][..., :7]).max() > 0  # scale and quaternion features did receive gradient


@pytest.mark.gpu
def test_hip_scale_rotation_form_at_scale_131072_gaussians():
    """PF3plat's native size: 2 source views x 256 x 256 pixel-aligned Gaussians as scale + quaternion records with per-view
    frames, 3 target views, colour + depth; image vs the oracle, gradient of the records vs the oracle chain."""
    from tests import gpu_util
    from pf3plat_amd.rasterizer import RasterConfig

    n, hw = 131072, (256, 256)
    sc = synthetic.make_scene(50, n, hw, num_views=3)
    g = torch.Generator().manual_seed(9)
    depth = sc.gaussians.means[0].norm(dim=-1)
    scales = (0.5 + 14.5 * torch.rand((1, n, 3), generator=g)) * depth[None, :, None] * (0.2 / (0.86 * 256))
    quats = torch.randn((1, n, 4), generator=g)
    frames = torch.linalg.qr(torch.randn(1, 2, 3, 3, generator=g))[0]
    records = torch.cat((scales, quats), -1)
    means, _, opac, colors = gpu_util.scene_tensors(sc)
    vb = gpu_util.scene_viewbuf(sc)
    cfg = RasterConfig(3, 1, 3, n, 256, 256, 4, 25, 4, True, 1 << 4, True)
    rng = np.random.default_rng(1)
    gc = torch.tensor(rng.uniform(0, 1, (3, 3, 256, 256)).astype(np.float32))
    ge = torch.tensor(rng.uniform(0, 1, (3, 256, 256)).astype(np.float32))
    dev = torch.device("cuda:0")
    hip = pf3plat_amd.rasterizer.HipBackend()
    a = [x.to(dev).contiguous() for x in (means, records, opac, colors)]
    hc, he, hr, saved = hip.forward(cfg, vb.to(dev), *a, None, frames=frames.to(dev))
    hg = hip.backward(cfg, saved, vb.to(dev), *a, None, gc.to(dev), ge.to(dev), False, frames=frames.to(dev))
    from tests.oracle_backend import OracleBackend
    ob = OracleBackend(threads=8)
    oc, oe, orad, osaved = ob.forward(cfg, vb, means, records, opac, colors, None, frames=frames)
    og = ob.backward(cfg, osaved, vb, means, records, opac, colors, None, gc, ge, False, frames=frames)
    assert rel_l2(hc.cpu().numpy(), oc.numpy()) < 1e-4 and rel_l2(he.cpu().numpy(), oe.numpy()) < 1e-4
    assert int((hr.cpu() != orad).sum()) <= 4
    for k, name in ((0, "means"), (1, "scale_rot"), (2, "opacities"), (3, "colors")):
        assert rel_l2(hg[k].cpu().numpy(), og[k].numpy()) < 1e-4, name
