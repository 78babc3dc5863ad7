"""SURVEY.md 8f-3 (opt-in camera-pose gradients): host chain rule and wrappers on CPU (oracle-backed backend); the raster
part itself is pinned by fp64 finite differences in tests/test_oracle_pose_grad.py and compared on the GPU in
tests/test_gpu_parity.py::test_camera_gradients_*."""
import numpy as np
import torch

import pf3plat_amd
from pf3plat_amd import synthetic
from pf3plat_amd.rasterizer import _SetupViews, views_from_cameras
from pf3plat_amd.types import Gaussians
from tests.util import rel_l2


def _records_fp64(ext, tan_x, tan_y, near, far, scale_invariant=True):
    """The record fields a pose reaches, as differentiable fp64 torch (cuda_splatting.py:64-71, :80-87 restated)."""
    scale = 1.0 / near if scale_invariant else torch.ones_like(near)
    top = torch.cat((ext[:, :3, :3], ext[:, :3, 3:] * scale[:, None, None]), dim=2)
    e = torch.cat((top, ext[:, 3:, :]), dim=1)
    n, f = near * scale, far * scale
    proj = torch.zeros((ext.shape[0], 4, 4), dtype=torch.float64)
    proj[:, 0, 0], proj[:, 1, 1] = 1 / tan_x, 1 / tan_y
    proj[:, 2, 2], proj[:, 2, 3], proj[:, 3, 2] = f / (f - n), -(f * n) / (f - n), 1.0
    view = torch.linalg.inv(e).transpose(1, 2)
    full = view @ proj.transpose(1, 2)
    return torch.cat((view.reshape(-1, 16), full.reshape(-1, 16), e[:, :3, 3]), dim=1)


def test_setup_views_backward_is_the_chain_rule_of_the_records(oracle_backend):
    sc = synthetic.make_scene(21, 10, (16, 16), num_views=3, near=1.7)
    ext = sc.extrinsics[0].double().requires_grad_(True)
    vb = views_from_cameras(ext, sc.intrinsics[0], sc.near[0], sc.far[0], torch.zeros(3), True, pose_gradients=True)
    assert vb.requires_grad and vb.shape == (3, 48)
    g = torch.randn((3, 48), generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    (d_ext,) = torch.autograd.grad(vb, ext, g.to(vb.dtype))
    ext2 = sc.extrinsics[0].double().requires_grad_(True)
    rec = _records_fp64(ext2, vb[:, 35].detach().double(), vb[:, 36].detach().double(), sc.near[0].double(), sc.far[0].double())
    assert rel_l2(rec.detach().numpy(), vb[:, :35].detach().numpy()) < 1e-6
    (d_ref,) = torch.autograd.grad(rec, ext2, g[:, :35])
    assert rel_l2(d_ext.numpy(), d_ref.numpy()) < 1e-5
    # off by default, and without a gradient request the records are plain tensors
    assert not views_from_cameras(ext, sc.intrinsics[0], sc.near[0], sc.far[0], torch.zeros(3)).requires_grad


def test_decoder_pose_gradients_reach_the_extrinsics(oracle_backend):
    sc = synthetic.make_scene(22, 120, (16, 24), num_views=2, near=1.5)
    dec = pf3plat_amd.DecoderSplattingCUDA(dataset_cfg=pf3plat_amd.decoder.DatasetCfgLike((0.2, 0.1, 0.0)))
    g = sc.gaussians
    w = torch.rand((1, 2, 3, 16, 24), generator=torch.Generator().manual_seed(1))
    wd = torch.rand((1, 2, 16, 24), generator=torch.Generator().manual_seed(2))
    ext = sc.extrinsics.clone().requires_grad_(True)
    means = g.means.clone().requires_grad_(True)
    out = dec.forward(Gaussians(means, g.covariances, g.harmonics, g.opacities), ext, sc.intrinsics, sc.near, sc.far, (16, 24),
                      depth_mode="depth", pose_gradients=True)
    ((out.color * w).sum() + (out.depth * wd).sum()).backward()
    assert ext.grad is not None and ext.grad.shape == (1, 2, 4, 4) and torch.isfinite(ext.grad).all()
    assert ext.grad[..., :3, :].abs().min() > 0
    # same call without the option: the reference's behaviour.  Its rasterizer gives the cameras nothing - a colour-only render
    # leaves extrinsics.grad at None - and its depth render reaches them only through extrinsics.inverse() in the fake colour
    # (cuda_splatting.py:239-242; pinned in tests/test_wrappers_cpu.py); the Gaussians get the same gradient either way
    ext_b = sc.extrinsics.clone().requires_grad_(True)
    means_b = g.means.clone().requires_grad_(True)
    out_b = dec.forward(Gaussians(means_b, g.covariances, g.harmonics, g.opacities), ext_b, sc.intrinsics, sc.near, sc.far, (16, 24))
    (out_b.color * w).sum().backward()
    assert ext_b.grad is None and out_b.depth is None and torch.equal(out.color, out_b.color)
    out_c = dec.forward(Gaussians(means_b, g.covariances, g.harmonics, g.opacities), ext_b, sc.intrinsics, sc.near, sc.far,
                        (16, 24), depth_mode="depth")
    means_b.grad = None
    ((out_c.color * w).sum() + (out_c.depth * wd).sum()).backward()
    assert ext_b.grad is not None and ext_b.grad.abs().sum() > 0 and not torch.allclose(ext_b.grad, ext.grad)
    assert torch.equal(out.color, out_c.color) and rel_l2(means_b.grad.numpy(), means.grad.numpy()) < 2e-5
    # rigid consistency: moving every Gaussian by +d (world) is the same as moving both cameras by -d, so the loss
    # gradient w.r.t. the camera centres sums to minus the gradient w.r.t. the means (fp32 oracle: loose)
    d_centres = ext.grad[0, :, :3, 3].sum(0)
    d_means = means.grad[0].sum(0)
    assert np.allclose(d_centres.numpy(), -d_means.numpy(), rtol=2e-3, atol=2e-3 * float(d_means.abs().max()))
