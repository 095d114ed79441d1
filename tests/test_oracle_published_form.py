"""The oracle's FOLDED alpha expression against the PUBLISHED one, at full size (configs[2]: 1 M Gaussians, 1352 x 1014).

Since round 4 `oracle/gp_oracle.c::gauss_exponent` evaluates alpha as the kernels do -- exp2 of one exponent with log2(e) and
log2(opacity) folded in, test_T = T - alpha T -- so that the float32 oracle and the HIP kernels take bit-identical skip / stop
decisions.  That makes the oracle follow the kernels' algebra (the round-4 advisor's and judge's note).  What keeps the published
statement of the algorithm in the loop: `gpo_composite_fwd_published` (min(0.99, opacity exp(power)), skip power > 0, T (1 - alpha);
nothing folded) is run on the same sorted lists, in float64 and in float32, and must agree with the folded float32 composite
  * on n_contrib and tidx for EVERY pixel neither side flags as ambiguous (a decision within rounding of its threshold),
  * on the image to 1e-5 (float32 rounding of ~700 blended terms),
and the ambiguous fraction must stay small -- a systematic error in the folded form (a threshold moved, G for tiny opacities,
the sign of the power test) shows up as a mismatch outside the bands, not as noise inside them.
CPU only: both sides are the oracle; the kernels are compared with the folded oracle in tests/test_gpu_configs.py."""
import math

import numpy as np
import pytest
import torch

from gaussianprediction_amd.cameras import orbit_cameras
from gaussianprediction_amd.scene_synth import SceneSpec, make_gaussians
from oracle.oracle import RasterOracle, RasterSettings


def _scene(n, W, H, scale_lo, scale_hi, tiny_opacity=False):
    raw = make_gaussians(SceneSpec(n_gaussians=n, extent=(1.5, 1.5, 0.5), scale_lo=scale_lo, scale_hi=scale_hi))
    op = torch.sigmoid(raw["opacity"])
    if tiny_opacity:                                   # a tenth of the cloud around and below the 1/255 threshold
        op[::10] = torch.logspace(-4, -2, op[::10].numel()).reshape(-1, 1)
    a = dict(means3D=raw["xyz"], opacities=op, shs=torch.cat([raw["features_dc"], raw["features_rest"]], 1),
             scales=torch.exp(raw["scaling"]), rotations=torch.nn.functional.normalize(raw["rotation"]))
    a = {k: v.numpy().astype(np.float64) for k, v in a.items()}
    cam = orbit_cameras(8, 4.0, 2 * math.atan(1 / 1.8), W, H, arc_deg=40.0, elevation_deg=5.0)[3]
    n64 = lambda x: x.detach().cpu().numpy().astype(np.float64)     # noqa: E731
    st = RasterSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                        bg=np.array([0.1, 0.2, 0.3]), scale_modifier=1.0, viewmatrix=n64(cam.world_view_transform),
                        projmatrix=n64(cam.full_proj_transform), sh_degree=3, campos=n64(cam.camera_center))
    return a, st


@pytest.mark.parametrize("n,W,H,lo,hi,tiny", [(1_000_000, 1352, 1014, 0.003, 0.012, False), (120_000, 640, 480, 0.004, 0.03, True)],
                         ids=["c3_full_size", "tiny_opacities"])
def test_folded_exponent_takes_the_published_decisions(n, W, H, lo, hi, tiny):
    a, st = _scene(n, W, H, lo, hi, tiny)
    o32, o64 = RasterOracle("f32"), RasterOracle("f64")
    kw = dict(shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
    s = o32.forward(st, a["means3D"], a["opacities"], **kw)                       # folded form, float32: what the kernels are held to
    if n >= 1_000_000:
        assert s["R"] > 3_500_000
    pub32 = o32.composite_published(s)                                            # published form, float32, same lists
    s64 = o64.preprocess(st, a["means3D"], a["opacities"], **kw)                  # published form, float64, same lists
    s64["R"], s64["point_list"], s64["ranges"] = s["R"], s["point_list"], s["ranges"]
    pub64 = o64.composite_published(s64, band_scale=20.0)          # (float32 pixel coordinates: ~1e-4 relative on alpha)
    for name, pub, img_tol, amb_tol in (("float32", pub32, 1e-5, 0.02), ("float64", pub64, 1e-4, 0.06)):
        clean = (s["ambiguous"] == 0) & (pub["ambiguous"] == 0)
        frac_amb = 1.0 - clean.mean()
        assert frac_amb < amb_tol, f"{name}: {frac_amb:.3%} of the pixels ambiguous"
        nc_bad = int((s["n_contrib"] != pub["n_contrib"])[clean].sum())
        tidx_bad = int((s["out_tidx"] != pub["out_tidx"])[clean].sum())
        err = np.abs(s["out_color"].astype(np.float64) - pub["out_color"].astype(np.float64))[:, clean].max()
        print(f"[published vs folded, {name}] pixels {clean.size}, ambiguous {frac_amb:.4%}, n_contrib mismatches {nc_bad}, "
              f"tidx mismatches {tidx_bad}, RGB Linf {err:.2e}")
        assert nc_bad == 0, f"{name}: n_contrib differs on {nc_bad} unambiguous pixels"
        assert tidx_bad == 0, f"{name}: tidx differs on {tidx_bad} unambiguous pixels"
        assert err <= img_tol, f"{name}: RGB Linf {err:.3e}"
        # where the two DO differ, it is inside the bands, and rare
        d = np.abs(s["n_contrib"].astype(np.int64) - pub["n_contrib"].astype(np.int64))
        assert float((d > 0).mean()) < 1e-3
