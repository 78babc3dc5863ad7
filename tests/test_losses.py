"""image losses.  tests/golden/loss_fixtures.npz holds values and autograd gradients of the REFERENCE's own `ssim`
(src/loss/loss_multissim.py), of LossMse's arithmetic and `compute_psnr` for seeded images (tests/golden/make_loss_fixtures.py
imports them on CPU in the build container).  CPU: the oracle's restatement against those; GPU: the one-launch HIP
evaluation (gsr_image_loss) against the fixtures and, on larger images, against the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import losses as oracle_losses
from tests.util import rel_l2

FIX = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_fixtures.npz"))
t = lambda k: torch.tensor(FIX[k])


@pytest.mark.parametrize("tag", ["A", "B", "C"])
def test_oracle_restatement_matches_reference_losses(tag):
    pred, target = t(tag + "_pred").requires_grad_(True), t(tag + "_target")
    s = oracle_losses.ssim_map(pred, target).mean()
    (gs,) = torch.autograd.grad(s, pred)
    np.testing.assert_allclose(s.item(), FIX[tag + "_ssim"], rtol=1e-6)
    assert rel_l2(gs.numpy(), FIX[tag + "_ssim_grad"]) < 1e-5
    loss, mse, _ = oracle_losses.photometric_loss(pred, target, 1.0, 0.0)
    (gm,) = torch.autograd.grad(loss, pred)
    np.testing.assert_allclose(mse.item(), FIX[tag + "_mse"], rtol=1e-6)
    np.testing.assert_allclose(gm.numpy(), FIX[tag + "_mse_grad"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(oracle_losses.psnr(target, pred.detach()).numpy(), FIX[tag + "_psnr"], rtol=1e-6)


def test_losses_refuse_cpu_tensors():
    from pf3plat_amd import losses

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        losses.photometric_loss(torch.zeros(1, 3, 8, 8), torch.zeros(1, 3, 8, 8))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["A", "B", "C"])
def test_hip_losses_match_reference_fixtures(tag):
    from pf3plat_amd import losses

    dev = "cuda:0"
    pred, target = t(tag + "_pred").to(dev).requires_grad_(True), t(tag + "_target").to(dev)
    s = losses.ssim(pred, target)
    np.testing.assert_allclose(s.item(), FIX[tag + "_ssim"], rtol=2e-5)
    (gs,) = torch.autograd.grad(s, pred)
    assert rel_l2(gs.cpu().numpy(), FIX[tag + "_ssim_grad"]) < 1e-4
    loss, mse, _ = losses.photometric_loss(pred, target, 1.0, 0.0)
    (gm,) = torch.autograd.grad(loss, pred)
    np.testing.assert_allclose(mse.item(), FIX[tag + "_mse"], rtol=1e-5)
    assert rel_l2(gm.cpu().numpy(), FIX[tag + "_mse_grad"]) < 1e-5
    np.testing.assert_allclose(losses.compute_psnr(target, pred.detach()).cpu().numpy(), FIX[tag + "_psnr"], rtol=1e-5)


@pytest.mark.gpu
def test_hip_combined_loss_on_rendered_size_images_and_loss_modules():
    """Both terms from one launch at the size the decoder renders (6 images of 256 x 256), gradient against the oracle; the
    LossMse / LossMultiSSIM modules slice the inner target views as the reference does."""
    from pf3plat_amd import losses
    from pf3plat_amd.types import DecoderOutput

    dev = "cuda:0"
    g = torch.Generator().manual_seed(4)
    target = torch.rand((6, 3, 256, 256), generator=g)
    pred = (target + 0.1 * torch.randn((6, 3, 256, 256), generator=g))
    p_gpu = pred.to(dev).requires_grad_(True)
    loss, mse, s = losses.photometric_loss(p_gpu, target.to(dev), 1.0, 0.25)
    loss.backward()
    p_cpu = pred.clone().requires_grad_(True)
    lo, mo, so = oracle_losses.photometric_loss(p_cpu, target, 1.0, 0.25)
    lo.backward()
    assert abs(loss.item() - lo.item()) < 1e-5 * abs(lo.item()) and abs(mse.item() - mo.item()) < 1e-5 * mo.item()
    assert abs(s.item() - so.item()) < 2e-5
    assert rel_l2(p_gpu.grad.cpu().numpy(), p_cpu.grad.numpy()) < 1e-4
    color = pred.reshape(1, 6, 3, 256, 256).to(dev).requires_grad_(True)
    batch = {"target": {"image": target.reshape(1, 6, 3, 256, 256).to(dev)}}
    out = DecoderOutput(color, None)
    l_mse = losses.LossMse(losses.LossMseCfg(2.0)).forward(out, batch, None, 0)
    l_ssim = losses.LossMultiSSIM(losses.LossMultiSSIMCfg(0.5)).forward(out, batch, None, 0)
    both = losses.LossPhotometric(losses.LossMseCfg(2.0), losses.LossMultiSSIMCfg(0.5)).forward(out, batch)
    ref_mse = 2.0 * ((pred[1:5] - target[1:5]) ** 2).mean().item()
    ref_ssim = 0.5 * (1 - oracle_losses.ssim_map(pred[1:5], target[1:5]).mean().item())
    assert abs(l_mse.item() - ref_mse) < 1e-5 * ref_mse and abs(l_ssim.item() - ref_ssim) < 1e-5
    assert abs(both.item() - (ref_mse + ref_ssim)) < 2e-5
    both.backward()
    assert torch.all(color.grad[:, 0] == 0) and torch.all(color.grad[:, -1] == 0) and color.grad[:, 1:-1].abs().sum() > 0
