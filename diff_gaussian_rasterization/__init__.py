"""Import-name shim: `from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`
[REF gaussian_renderer/__init__.py:14] resolves to the MI355X implementation."""
from gaussianprediction_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                               rasterize_gaussians)
