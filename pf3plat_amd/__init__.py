"""pf3plat_amd — MI355X-native differentiable 3D-Gaussian rasterizer behind PF3plat's
`cuda_splatting` operator surface (C ABI in include/gsr.h)."""
from .types import DecoderOutput, DepthRenderingMode, Gaussians  # noqa: F401
from .rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    RasterConfig,
    get_backend,
    pack_views,
    rasterize_views,
    views_from_cameras,
)
from .splatting import (  # noqa: F401
    render_cuda,
    render_cuda_orthographic,
    render_depth_cuda,
    render_views,
)
from .adapter import AdaptedGaussians, GaussianAdapter, GaussianAdapterCfg  # noqa: F401
from .decoder import DecoderSplattingCUDA, DecoderSplattingCUDACfg, get_decoder  # noqa: F401
from .geometry import depth_to_relative_disparity, get_fov, get_projection_matrix, homogenize_points  # noqa: F401

__version__ = "0.1.0"
from .ply_export import export_ply, gaussians_from_ply, read_ply  # noqa: F401,E402
from .sh_rotation import rotate_sh  # noqa: F401,E402
from . import losses  # noqa: F401,E402
