"""Decoder layer: counterpart of the reference's `DecoderSplattingCUDA`
(src/model/decoder/decoder_splatting_cuda.py:20-91) and its `Decoder` base / registry
(src/model/decoder/decoder.py:28-48, src/model/decoder/__init__.py:5-13).

Same constructor shape (a cfg with `name`, a dataset cfg carrying `background_color`), same
`forward(gaussians, extrinsics, intrinsics, near, far, image_shape, depth_mode) -> DecoderOutput`
and `render_depth(...)`.  `forward` goes through the fused `render_views` path: one launch chain
for all (batch, view) pairs, Gaussians shared between the views of a scene instead of
`repeat`-ed V times (:52-56), depth blended in the same pass instead of a second raster pass (:60-66).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Literal, Optional, Sequence

import torch
from torch import Tensor, nn

from .splatting import render_depth_cuda, render_views
from .types import DecoderOutput, DepthRenderingMode, Gaussians


@dataclass
class DecoderSplattingCUDACfg:
    name: Literal["splatting_cuda"] = "splatting_cuda"


@dataclass
class DatasetCfgLike:
    """The one dataset-config field the decoder reads (reference config/dataset/re10k.yaml:10)."""
    background_color: Sequence[float] = (0.0, 0.0, 0.0)


class Decoder(nn.Module):
    def __init__(self, cfg, dataset_cfg) -> None:
        super().__init__()
        self.cfg = cfg
        self.dataset_cfg = dataset_cfg


class DecoderSplattingCUDA(Decoder):
    background_color: Tensor

    def __init__(self, cfg: Optional[DecoderSplattingCUDACfg] = None, dataset_cfg=None, on_overflow: Optional[str] = "nan") -> None:
        """on_overflow: what the raster backend does when a deferred training forward outgrows its pair workspace.  "nan" (this
        decoder's default - it is PF3plat's training surface, and PF3plat's optimizer step skips a step whose gradients hold a NaN,
        src/model/model_wrapper.py:224-238): the step's image, loss and gradients are NaN, a `RasterOverflowWarning` is issued,
        nothing is raised - every DDP rank still enters the gradient all-reduce.  "raise" (the library's own default, for callers
        with a plain `optimizer.step()`): RuntimeError from that step's backward.  None: leave the backend as it is."""
        cfg = cfg if cfg is not None else DecoderSplattingCUDACfg()
        dataset_cfg = dataset_cfg if dataset_cfg is not None else DatasetCfgLike()
        super().__init__(cfg, dataset_cfg)
        if on_overflow is not None:
            from .rasterizer import get_backend

            backend = get_backend()
            if hasattr(backend, "on_overflow"):  # (tests slide other backend objects under the wrappers)
                backend.on_overflow = on_overflow
        self.register_buffer(
            "background_color",
            torch.tensor(list(dataset_cfg.background_color), dtype=torch.float32),
            persistent=False,
        )

    def forward(
        self,
        gaussians: Gaussians,
        extrinsics: Tensor,  # (batch, view, 4, 4)
        intrinsics: Tensor,  # (batch, view, 3, 3)
        near: Tensor,  # (batch, view)
        far: Tensor,  # (batch, view)
        image_shape: tuple,
        depth_mode: Optional[DepthRenderingMode] = None,
        pose_gradients: bool = False,  # extension: let the render's gradient reach `extrinsics`
    ) -> DecoderOutput:
        color, depth = render_views(
            extrinsics, intrinsics, near, far, image_shape,
            self.background_color.to(extrinsics.device),
            gaussians.means, gaussians.covariances, gaussians.harmonics, gaussians.opacities,
            depth_mode=depth_mode, gaussian_scales=gaussians.scales, gaussian_rotations=gaussians.rotations, frames=gaussians.frames,
            pose_gradients=pose_gradients,
        )
        return DecoderOutput(color, depth)

    def render_depth(
        self,
        gaussians: Gaussians,
        extrinsics: Tensor,
        intrinsics: Tensor,
        near: Tensor,
        far: Tensor,
        image_shape: tuple,
        mode: DepthRenderingMode = "depth",
    ) -> Tensor:  # (batch, view, height, width)
        b, v, _, _ = extrinsics.shape
        # the v views of a scene share the scene's one copy of the Gaussians (the reference repeats them v times, :79-86)
        result = render_depth_cuda(
            extrinsics.reshape(b * v, 4, 4), intrinsics.reshape(b * v, 3, 3), near.reshape(b * v), far.reshape(b * v),
            image_shape, gaussians.means, gaussians.covariances, gaussians.opacities, mode=mode,
            gaussian_scales=gaussians.scales, gaussian_rotations=gaussians.rotations, frames=gaussians.frames,
        )
        h, w = image_shape
        return result.reshape(b, v, h, w)


DECODERS = {"splatting_cuda": DecoderSplattingCUDA}


def get_decoder(decoder_cfg, dataset_cfg) -> Decoder:
    return DECODERS[decoder_cfg.name](decoder_cfg, dataset_cfg)
