"""Data contracts of the hot path (reference src/model/types.py:7-18, decoder/decoder.py:11-22)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Literal, Optional

from torch import Tensor

DepthRenderingMode = Literal["depth", "log", "disparity", "relative_disparity"]


@dataclass
class Gaussians:
    means: Tensor  # (batch, gaussian, 3)
    covariances: Tensor  # (batch, gaussian, 3, 3) full symmetric world-space
    harmonics: Tensor  # (batch, gaussian, 3, d_sh)
    opacities: Tensor  # (batch, gaussian), already in (0, 1)

    def clone(self) -> "Gaussians":
        return Gaussians(
            means=self.means.clone(),
            covariances=self.covariances.clone(),
            harmonics=self.harmonics.clone(),
            opacities=self.opacities.clone(),
        )


@dataclass
class DecoderOutput:
    color: Tensor  # (batch, view, 3, height, width)
    depth: Optional[Tensor]  # (batch, view, height, width) or None
