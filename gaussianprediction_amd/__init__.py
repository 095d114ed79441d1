"""gaussianprediction_amd -- MI355X (gfx950) native dynamic-Gaussian render hot path of
BoMingZhao/GaussianPrediction: deformation MLP, projection, tile binning / radix sort and the
alpha-composite forward + backward as hand-written HIP kernels behind the reference's own
`gaussian_renderer.render()` / `diff_gaussian_rasterization.GaussianRasterizer` API."""
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401
from .renderer import render, render_motion, SpeculativeRenderer  # noqa: F401
from .gaussian_model import GaussianModel  # noqa: F401
from .deformable_field import Deformable_Field  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "render", "render_motion", "SpeculativeRenderer", "GaussianModel",
           "Deformable_Field"]
