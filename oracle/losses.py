"""Oracle side of the image losses: the reference's photometric terms restated with CPU torch ops, their
gradient by autograd.  TEST INFRASTRUCTURE - only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this; the product evaluates them in one HIP launch (gsr_image_loss) and is compared with these functions by the tests.

Pinned to the reference by tests/golden/loss_fixtures.npz (tests/test_losses.py): values and gradients recorded from the
reference's own `ssim`, LossMse arithmetic and `compute_psnr` for seeded images.

Follows (file:line under /root/reference):
  * src/loss/loss_mse.py:35-36            weight x mean((prediction - target)^2)
  * src/loss/loss_multissim.py:41-83      1 - mean SSIM map; 11 x 11 window = outer product of a normalised Gaussian (sigma 1.5),
                                           depthwise convolutions with zero padding 5, C1 = 0.01^2, C2 = 0.03^2
  * src/evaluation/metrics.py:11-19       psnr = -10 log10 mean((clip(gt) - clip(pred))^2) per image
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import Tensor


def ssim_window(dtype=torch.float32) -> Tensor:
    g = torch.tensor([math.exp(-((k - 5) ** 2) / (2 * 1.5 ** 2)) for k in range(11)], dtype=torch.float32)
    g = g / g.sum()
    return (g[:, None] @ g[None, :]).to(dtype)


def ssim_map(a: Tensor, b: Tensor) -> Tensor:
    """(n, c, h, w) x 2 -> SSIM map (n, c, h, w)."""
    c = a.shape[1]
    w = ssim_window(a.dtype)[None, None].expand(c, 1, 11, 11)
    blur = lambda t: F.conv2d(t, w, padding=5, groups=c)
    mu_a, mu_b = blur(a), blur(b)
    var_a, var_b, cov = blur(a * a) - mu_a * mu_a, blur(b * b) - mu_b * mu_b, blur(a * b) - mu_a * mu_b
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu_a * mu_b + c1) * (2 * cov + c2)) / ((mu_a * mu_a + mu_b * mu_b + c1) * (var_a + var_b + c2))


def photometric_loss(prediction: Tensor, target: Tensor, mse_weight: float, ssim_weight: float):
    """-> (loss, mse, mean ssim): loss = mse_weight * mse + ssim_weight * (1 - mean ssim); differentiable in `prediction`."""
    mse = ((prediction - target) ** 2).mean()
    s = ssim_map(prediction, target).mean()
    return mse_weight * mse + ssim_weight * (1 - s), mse, s


def psnr(ground_truth: Tensor, predicted: Tensor) -> Tensor:
    d = ground_truth.clamp(0, 1) - predicted.clamp(0, 1)
    return -10 * (d * d).flatten(1).mean(1).log10()
