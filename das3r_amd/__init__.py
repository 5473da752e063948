"""das3r_amd — MI355X-native (gfx950) differentiable Gaussian-splat rasterizer + distCUDA2 for DAS3R.

Only the hot path lives here: csrc/ (hand-written HIP kernels + the C-ABI of include/das3r_raster.h) and the
host-side mirrors of the reference's Python interface for that path.  Drop-in module names are provided by the
top-level `diff_gaussian_rasterization` and `simple_knn` packages.
"""
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians  # noqa: F401
from .knn import distCUDA2  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "distCUDA2"]
