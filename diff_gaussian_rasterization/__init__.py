"""Drop-in module name for the reference's import at src/model/decoder/cuda_splatting.py:5-8:

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

With this repository on PYTHONPATH the reference's decoder resolves the operator to the MI355X
implementation unchanged.
"""
from pf3plat_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
