#!/usr/bin/env python
"""The composite kernels at three splat densities (round-2 verdict item 4: "report composite fwd/bwd ms and roofline fractions at
R/N ~ 4, 9, 16 so the kernels are shown not to be tuned to one density").  configs[2]'s scene and camera, splat scales x 1,
x 1.75, x 2.9; rasterizer forward + backward only; algorithmic bytes = SURVEY 8d's formulas.  One JSON line per density."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianprediction_amd import _lib
from gaussianprediction_amd.cameras import orbit_cameras
from gaussianprediction_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, raster_forward_debug
from gaussianprediction_amd.scene_synth import SceneSpec, make_gaussians

N, W, H = 1_000_000, 1352, 1014
dev = torch.device("cuda", 0)
cam = orbit_cameras(8, 4.0, 2 * math.atan(1 / 1.8), W, H, arc_deg=40.0, elevation_deg=5.0, device=dev)[3]
rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                                   bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
                                   projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center, prefiltered=False)
T, P = ((W + 15) // 16) * ((H + 15) // 16), W * H
gw = torch.randn(3, H, W, device=dev)
for mult in (1.0, 1.75, 2.9):
    raw = make_gaussians(SceneSpec(n_gaussians=N, extent=(1.5, 1.5, 0.5), scale_lo=0.003 * mult, scale_hi=0.012 * mult), device=dev)
    x = dict(means3D=raw["xyz"], opacities=torch.sigmoid(raw["opacity"]), shs=torch.cat([raw["features_dc"], raw["features_rest"]], 1).contiguous(),
             scales=torch.exp(raw["scaling"]), rotations=torch.nn.functional.normalize(raw["rotation"]))
    with torch.no_grad():
        dbg = raster_forward_debug(rs, x["means3D"], x["opacities"], shs=x["shs"], scales=x["scales"], rotations=x["rotations"])
    R, nvis = dbg["R"], int((dbg["radii"] > 0).sum())
    rast = GaussianRasterizer(raster_settings=rs)
    m2 = torch.zeros(N, 3, device=dev, requires_grad=True)

    def once():
        L = {k: v.detach().requires_grad_(True) for k, v in x.items()}
        img, _, _, _ = rast(means3D=L["means3D"], means2D=m2, opacities=L["opacities"], shs=L["shs"], scales=L["scales"], rotations=L["rotations"])
        (img * gw).sum().backward()

    for _ in range(3):
        once()
    torch.cuda.synchronize()
    _lib.profile_enable(2); _lib.profile_collect()
    for _ in range(10):
        once()
    torch.cuda.synchronize()
    prof = _lib.profile_collect(); _lib.profile_enable(0)
    ms = {k: v[1] / v[0] for k, v in prof.items()}
    bf, bb = 44 * R + 8 * T + 28 * P, 44 * R + 8 * T + 24 * P + 40 * nvis
    lens = (dbg["ranges"][:, 1] - dbg["ranges"][:, 0]).float()
    print(json.dumps({"scale_multiplier": mult, "R": R, "R_per_gaussian": round(R / N, 2), "visible": nvis,
                      "tile_list_mean": round(float(lens.mean()), 1), "tile_list_max": int(lens.max()),
                      "composite_fwd_ms": round(ms["composite_fwd"], 4), "composite_fwd_GBps": round(bf / ms["composite_fwd"] / 1e6, 1),
                      "composite_fwd_frac_of_8TBps": round(bf / ms["composite_fwd"] / 1e6 / 8000, 4),
                      "composite_bwd_ms": round(ms["composite_bwd"], 4), "composite_bwd_GBps": round(bb / ms["composite_bwd"] / 1e6, 1),
                      "composite_bwd_frac_of_8TBps": round(bb / ms["composite_bwd"] / 1e6 / 8000, 4),
                      "depth_sort_ms": round(ms["depth_sort"], 4), "tile_sort_ms": round(ms["tile_sort"], 4),
                      "duplicate_ms": round(ms["duplicate"], 4), "preprocess_fwd_ms": round(ms["preprocess_fwd"], 4),
                      "preprocess_bwd_ms": round(ms["preprocess_bwd"], 4)}), flush=True)
