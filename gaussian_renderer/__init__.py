"""Import-name shim: `from gaussian_renderer import render, render_motion, GaussianModel`
[REF train.py:20, eval.py:21-22,31] resolves to the MI355X implementation."""
from gaussianprediction_amd.gaussian_model import GaussianModel  # noqa: F401
from gaussianprediction_amd.renderer import render, render_motion  # noqa: F401
