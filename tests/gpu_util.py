"""Helpers shared by the -m gpu tests: run the HIP path and the oracle on the same seeded inputs."""
import numpy as np
import torch

import gaussianprediction_amd as gpa
from gaussianprediction_amd.rasterizer import raster_forward_debug
from oracle.oracle import RasterOracle
from util import np64


def torch_settings(st, device="cuda"):
    f = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=device)
    return gpa.GaussianRasterizationSettings(
        image_height=st.image_height, image_width=st.image_width, tanfovx=st.tanfovx, tanfovy=st.tanfovy, bg=f(st.bg),
        scale_modifier=st.scale_modifier, viewmatrix=f(st.viewmatrix), projmatrix=f(st.projmatrix),
        sh_degree=st.sh_degree, campos=f(st.campos), prefiltered=False)


def f32_settings(st):
    """Round the camera through float32 so oracle and HIP see bit-identical inputs."""
    import copy
    s = copy.copy(st)
    for k in ("bg", "viewmatrix", "projmatrix", "campos"):
        setattr(s, k, np.asarray(getattr(st, k), np.float32).astype(np.float64))
    s.tanfovx = float(np.float32(st.tanfovx))
    s.tanfovy = float(np.float32(st.tanfovy))
    return s


def scene_to_device(scene, device="cuda", requires_grad=False):
    return {k: v.to(torch.float32).to(device).requires_grad_(requires_grad) for k, v in scene.items()}


def scene_f32_numpy(scene):
    return {k: v.to(torch.float32).numpy().astype(np.float64) for k, v in scene.items()}


def hip_forward_debug(st, scene_dev, **kw):
    rs = torch_settings(st)
    return raster_forward_debug(rs, scene_dev["means3D"], scene_dev["opacities"], shs=kw.get("shs", scene_dev.get("shs")),
                                colors_precomp=kw.get("colors_precomp"), scales=kw.get("scales", scene_dev.get("scales")),
                                rotations=kw.get("rotations", scene_dev.get("rotations")), cov3D_precomp=kw.get("cov3D_precomp"))
