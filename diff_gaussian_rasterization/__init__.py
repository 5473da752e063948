"""Drop-in module name for DAS3R: `from diff_gaussian_rasterization import GaussianRasterizationSettings,
GaussianRasterizer` (/root/reference/gaussian_renderer/__init__.py:14-17) resolves to the MI355X implementation."""
from das3r_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians,  # noqa: F401
                                  _RasterizeGaussians, cpu_deep_copy_tuple)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
