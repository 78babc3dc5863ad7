"""Data contracts of the hot path: what the decoder receives and returns.

The field names and tensor shapes are the interface PF3plat's encoder, decoder and losses share (reference
src/model/types.py:7-18 for the Gaussian set, src/model/decoder/decoder.py:11-22 for the decoder's output and the depth
modes); everything else here (shape checks, device moves, derived sizes) is this package's own.
"""
from __future__ import annotations

import dataclasses
from math import isqrt
from typing import Callable, Literal, Optional

from torch import Tensor

#: how `render_depth` / `DecoderSplattingCUDA.forward(depth_mode=...)` map camera-space z before blending it
DepthRenderingMode = Literal["depth", "log", "disparity", "relative_disparity"]


def _map_tensors(obj, fn: Callable[[Tensor], Tensor]):
    """A copy of a dataclass instance with `fn` applied to every tensor field (None fields stay None)."""
    values = {f.name: getattr(obj, f.name) for f in dataclasses.fields(obj)}
    return type(obj)(**{k: (fn(v) if isinstance(v, Tensor) else v) for k, v in values.items()})


@dataclasses.dataclass
class Gaussians:
    """One set of 3D Gaussians per scene of the batch.

    means        (scene, gaussian, 3)      world-space centres
    covariances  (scene, gaussian, 3, 3)   full symmetric world-space covariance (only the upper triangle is read)
    harmonics    (scene, gaussian, 3, d_sh) SH coefficients per colour channel, d_sh = (degree + 1)^2
    opacities    (scene, gaussian)         already in (0, 1)
    """

    means: Tensor
    covariances: Tensor
    harmonics: Tensor
    opacities: Tensor

    # -- derived sizes
    @property
    def num_scenes(self) -> int:
        return int(self.means.shape[0])

    @property
    def num_gaussians(self) -> int:
        return int(self.means.shape[1])

    @property
    def d_sh(self) -> int:
        return int(self.harmonics.shape[-1])

    @property
    def sh_degree(self) -> int:
        return isqrt(self.d_sh) - 1

    # -- copies
    def clone(self) -> "Gaussians":
        return _map_tensors(self, lambda t: t.clone())

    def detach(self) -> "Gaussians":
        return _map_tensors(self, lambda t: t.detach())

    def to(self, *args, **kwargs) -> "Gaussians":
        return _map_tensors(self, lambda t: t.to(*args, **kwargs))

    def check(self) -> "Gaussians":
        """Raise ValueError unless the four tensors agree on (scene, gaussian) and have the documented trailing shapes."""
        s, g = self.means.shape[:2]
        want = {"means": (s, g, 3), "covariances": (s, g, 3, 3), "opacities": (s, g)}
        for name, shape in want.items():
            if tuple(getattr(self, name).shape) != shape:
                raise ValueError(f"Gaussians.{name}: expected shape {shape}, got {tuple(getattr(self, name).shape)}")
        h = tuple(self.harmonics.shape)
        if len(h) != 4 or h[:3] != (s, g, 3) or (isqrt(h[3]) ** 2 != h[3]):
            raise ValueError(f"Gaussians.harmonics: expected shape ({s}, {g}, 3, (degree + 1)^2), got {h}")
        return self


@dataclasses.dataclass
class DecoderOutput:
    """color (scene, view, 3, height, width); depth (scene, view, height, width) or None when no depth mode was asked for."""

    color: Tensor
    depth: Optional[Tensor]
