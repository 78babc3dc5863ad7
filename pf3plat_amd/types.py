"""Data contracts of the hot path: what the decoder receives and returns (field names and shapes are the interface
PF3plat's encoder, decoder and losses share: reference src/model/types.py:7-18, src/model/decoder/decoder.py:11-22)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Literal, Optional

from torch import Tensor

#: how `render_depth` / `DecoderSplattingCUDA.forward(depth_mode=...)` map camera-space z before blending it
DepthRenderingMode = Literal["depth", "log", "disparity", "relative_disparity"]


@dataclass
class Gaussians:
    means: Tensor  # (scene, gaussian, 3) world-space centres
    covariances: Optional[Tensor]  # (scene, gaussian, 3, 3) symmetric world-space covariance (the upper triangle is what is read)
    harmonics: Tensor  # (scene, gaussian, 3, d_sh) SH coefficients per colour channel, d_sh = (degree + 1)^2
    opacities: Tensor  # (scene, gaussian) in (0, 1)
    # the form the encoder's adapter produces (pf3plat_amd.adapter): when `covariances` is None the decoder hands these to the
    # raster kernels, which build Sigma = (F R) diag(scale^2) (F R)^T on load
    scales: Optional[Tensor] = None  # (scene, gaussian, 3)
    rotations: Optional[Tensor] = None  # (scene, gaussian, 4) quaternions x, y, z, w
    frames: Optional[Tensor] = None  # (scene, F, 3, 3) world rotation of each of the F equal consecutive groups of Gaussians

    def clone(self) -> "Gaussians":
        """A deep copy (reference src/model/types.py:12-18), the optional fields included when present."""
        c = lambda t: None if t is None else t.clone()
        return Gaussians(means=self.means.clone(), covariances=c(self.covariances), harmonics=self.harmonics.clone(),
                         opacities=self.opacities.clone(), scales=c(self.scales), rotations=c(self.rotations), frames=c(self.frames))


@dataclass
class DecoderOutput:
    color: Tensor  # (scene, view, 3, height, width)
    depth: Optional[Tensor]  # (scene, view, height, width), or None when no depth mode was asked for
