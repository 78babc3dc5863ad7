"""Photometric losses next to the raster path: counterparts of the reference's `LossMse`
(src/loss/loss_mse.py:23-36), `LossMultiSSIM` / `ssim` (src/loss/loss_multissim.py:24-83) and `compute_psnr`
(src/evaluation/metrics.py:11-19), all evaluated by ONE launch of the raster library (`gsr_image_loss`) that also writes
dL/dprediction in the layout the rasterizer's backward reads - the loss's backward is then a no-op (the gradient already
exists) instead of five depthwise convolutions and their transposes.  No CPU fallback: tensors must be on a ROCm device.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
from torch import Tensor

from . import _lib
from .rasterizer import _on_device, _stream_ptr


def _launch(prediction: Tensor, target: Tensor, mse_weight: float, ssim_weight: float, want_grad: bool):
    """-> (sums (n, 4): per image squared error, clipped squared error, SSIM map, 0; totals (4,): loss, mse, mean ssim, 0;
    dL/dprediction or None; elements per image).  Two launches (gsr_image_loss, gsr_image_loss_finish), no torch op."""
    if not (prediction.is_cuda and target.is_cuda):
        raise RuntimeError("pf3plat_amd losses: tensors must be on a ROCm device (there is no CPU fallback path)")
    if prediction.shape != target.shape or prediction.dim() != 4 or prediction.shape[1] != 3:
        raise ValueError(f"expected two (n, 3, h, w) images, got {tuple(prediction.shape)} and {tuple(target.shape)}")
    lib = _lib.load()
    f32 = torch.float32
    pred = prediction.detach()
    tgt = target.detach()
    if pred.dtype != f32 or not pred.is_contiguous():
        pred = pred.to(f32).contiguous()
    if tgt.dtype != f32 or not tgt.is_contiguous():
        tgt = tgt.to(f32).contiguous()
    n, _, h, w = pred.shape
    dev = pred.device
    if n == 0:
        z = torch.zeros((1, 4), dtype=f32, device=dev)
        return z[:0], z[0], (torch.empty_like(pred) if want_grad else None), 3 * h * w
    slots = int(lib.gsr_image_loss_partials(n, h, w))
    partials = torch.empty((slots, 4), dtype=f32, device=dev)
    out = torch.empty((n + 1, 4), dtype=f32, device=dev)  # the images' sums, then the batch's totals
    grad = torch.empty_like(pred) if want_grad else None
    stream = _stream_ptr(dev)
    pp = partials.data_ptr()
    with _on_device(dev):
        rc = lib.gsr_image_loss(n, h, w, pred.data_ptr(), tgt.data_ptr(), float(mse_weight), float(ssim_weight),
                                None if grad is None else grad.data_ptr(), pp, stream)
        if rc == 0:
            rc = lib.gsr_image_loss_finish(n, h, w, pp, float(mse_weight), float(ssim_weight), out.data_ptr(), out[n].data_ptr(), stream)
    if rc != 0:
        raise RuntimeError(f"gsr_image_loss failed with code {rc}")
    return out[:n], out[n], grad, 3 * h * w


class _Photometric(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prediction, target, mse_weight, ssim_weight):
        _sums, totals, grad, _per_image = _launch(prediction, target, mse_weight, ssim_weight, prediction.requires_grad)
        ctx.grad = grad
        loss, mse, ssim = totals[0], totals[1], totals[2]
        ctx.mark_non_differentiable(mse, ssim)
        return loss, mse, ssim

    @staticmethod
    def backward(ctx, g_loss, _g_mse, _g_ssim):
        return (None if ctx.grad is None else ctx.grad * g_loss), None, None, None


def photometric_loss(prediction: Tensor, target: Tensor, mse_weight: float = 1.0, ssim_weight: float = 0.0):
    """prediction, target (n, 3, h, w) -> (loss, mse, mean ssim) with loss = mse_weight * mse + ssim_weight * (1 - mean ssim);
    differentiable in `prediction` (the gradient is produced by the same launch)."""
    return _Photometric.apply(prediction, target, float(mse_weight), float(ssim_weight))


def ssim(img1: Tensor, img2: Tensor) -> Tensor:
    """Mean SSIM map of two (n, 3, h, w) batches (the reference's `ssim(img1, img2)` with its defaults)."""
    return photometric_loss(img1, img2, 0.0, -1.0)[0] + 1.0  # loss = -(1 - ssim)


@torch.no_grad()
def compute_psnr(ground_truth: Tensor, predicted: Tensor) -> Tensor:
    """(batch, 3, h, w) x 2 -> (batch,): -10 log10 of the mean squared error of the inputs clipped to [0, 1]."""
    sums, _totals, _, per_image = _launch(predicted, ground_truth, 0.0, 0.0, False)
    return -10 * (sums[:, 1] / per_image).log10()


@dataclass
class LossMseCfg:
    weight: float


@dataclass
class LossMultiSSIMCfg:
    weight: float


def _inner_views(prediction_color: Tensor, batch) -> tuple:
    """The reference compares the target views without the first and the last one ([:, 1:-1]); (b, v, 3, h, w) -> (b v, 3, h, w)."""
    pred = prediction_color[:, 1:-1]
    tgt = batch["target"]["image"][:, 1:-1]
    return pred.reshape(-1, *pred.shape[2:]), tgt.reshape(-1, *tgt.shape[2:])


class LossMse(torch.nn.Module):
    """`forward(prediction, batch, ...)` as the reference's LossMse: weight x mean squared error over the inner target views."""

    def __init__(self, cfg: LossMseCfg):
        super().__init__()
        self.cfg = cfg

    def forward(self, prediction, batch, gaussians=None, global_step: int = 0, *unused) -> Tensor:
        return photometric_loss(*_inner_views(prediction.color, batch), self.cfg.weight, 0.0)[0]


class LossMultiSSIM(torch.nn.Module):
    """`forward(prediction, batch, ...)` as the reference's LossMultiSSIM: weight x (1 - mean SSIM) over the inner target views."""

    def __init__(self, cfg: LossMultiSSIMCfg):
        super().__init__()
        self.cfg = cfg

    def forward(self, prediction, batch, gaussians=None, global_step: int = 0, *unused) -> Tensor:
        return photometric_loss(*_inner_views(prediction.color, batch), 0.0, self.cfg.weight)[0]


class LossPhotometric(torch.nn.Module):
    """Both terms from one launch (what a training step that uses the two reference losses together should call)."""

    def __init__(self, mse: Optional[LossMseCfg] = None, ssim: Optional[LossMultiSSIMCfg] = None):
        super().__init__()
        self.mse_weight = 0.0 if mse is None else mse.weight
        self.ssim_weight = 0.0 if ssim is None else ssim.weight

    def forward(self, prediction, batch, *unused) -> Tensor:
        return photometric_loss(*_inner_views(prediction.color, batch), self.mse_weight, self.ssim_weight)[0]
