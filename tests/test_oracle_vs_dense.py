"""CPU: the oracle (C, float64 shadow) against an independent dense torch-autograd restatement.
This is what validates the oracle's hand-derived backward; the oracle's rasterizer arithmetic is
otherwise 'parity unpinned' (reference rasterizer source absent)."""
import numpy as np
import pytest
import torch

from dense_ref import dense_render
from oracle.oracle import RasterOracle
from util import np64, rel_l2, small_scene


def _run_dense(scene, st, use_colors=False, use_cov=False, with_depth=True, seed=0):
    leaves = {k: v.clone().requires_grad_(True) for k, v in scene.items()}
    N = leaves["means3D"].shape[0]
    m2d = torch.zeros(N, 3, dtype=torch.float64, requires_grad=True)
    colors = cov = None
    if use_colors:
        colors = torch.rand(N, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(3)).requires_grad_(True)
    if use_cov:
        from dense_ref import quat_R
        L = quat_R(scene["rotations"]) * scene["scales"][:, None, :]
        S = L @ L.transpose(1, 2)
        cov = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).detach().requires_grad_(True)
    img, depth, radii, tidx = dense_render(
        leaves["means3D"], m2d, None if use_colors else leaves["shs"], colors, leaves["opacities"],
        None if use_cov else leaves["scales"], None if use_cov else leaves["rotations"], cov,
        torch.tensor(st.viewmatrix), torch.tensor(st.projmatrix), torch.tensor(st.campos), torch.tensor(st.bg),
        st.image_height, st.image_width, st.tanfovx, st.tanfovy, st.sh_degree, st.scale_modifier)
    g = torch.Generator().manual_seed(seed)
    wimg = torch.randn(img.shape, dtype=torch.float64, generator=g)
    wdep = torch.randn(depth.shape, dtype=torch.float64, generator=g) * (0.1 if with_depth else 0.0)
    loss = (img * wimg).sum() + (depth * wdep).sum()
    loss.backward()
    return dict(img=img, depth=depth, radii=radii, tidx=tidx, wimg=wimg, wdep=wdep, leaves=leaves, m2d=m2d,
                colors=colors, cov=cov)


@pytest.mark.parametrize("sh_degree", [0, 1, 2, 3])
def test_oracle_f64_matches_dense_autograd(sh_degree):
    scene, st, _ = small_scene(n=220, W=70, H=50, seed=11 + sh_degree, sh_degree=sh_degree)
    d = _run_dense(scene, st)
    orc = RasterOracle("f64")
    s = orc.forward(st, np64(scene["means3D"]), np64(scene["opacities"]), shs=np64(scene["shs"]),
                    scales=np64(scene["scales"]), rotations=np64(scene["rotations"]))
    assert (s["radii"] > 0).sum() > 50
    np.testing.assert_array_equal(s["radii"], d["radii"].numpy())
    assert np.abs(s["out_color"] - np64(d["img"])).max() < 1e-10
    assert np.abs(s["out_depth"] - np64(d["depth"])).max() < 1e-10
    amb = s["ambiguous"] & 2
    assert ((s["out_tidx"] != d["tidx"].numpy()) & (amb == 0)).sum() == 0
    g = orc.backward(s, np64(d["wimg"]), np64(d["wdep"]))
    L = d["leaves"]
    assert rel_l2(g["means3D"], np64(L["means3D"].grad)) < 1e-9
    assert rel_l2(g["shs"], np64(L["shs"].grad)) < 1e-9
    assert rel_l2(g["opacities"], np64(L["opacities"].grad)) < 1e-9
    assert rel_l2(g["scales"], np64(L["scales"].grad)) < 1e-9
    assert rel_l2(g["rotations"], np64(L["rotations"].grad)) < 1e-9
    assert rel_l2(g["means2D"], np64(d["m2d"].grad[:, :2])) < 1e-9


def test_oracle_f64_precomputed_color_and_cov():
    scene, st, _ = small_scene(n=180, W=64, H=48, seed=5)
    d = _run_dense(scene, st, use_colors=True, use_cov=True)
    orc = RasterOracle("f64")
    s = orc.forward(st, np64(scene["means3D"]), np64(scene["opacities"]), colors_precomp=np64(d["colors"]),
                    cov3D_precomp=np64(d["cov"]))
    assert np.abs(s["out_color"] - np64(d["img"])).max() < 1e-10
    g = orc.backward(s, np64(d["wimg"]), np64(d["wdep"]))
    assert rel_l2(g["colors_precomp"], np64(d["colors"].grad)) < 1e-9
    assert rel_l2(g["cov3D_precomp"], np64(d["cov"].grad)) < 1e-9
    assert rel_l2(g["means3D"], np64(d["leaves"]["means3D"].grad)) < 1e-9


def test_oracle_f32_close_to_f64_and_edge_cases():
    scene, st, _ = small_scene(n=400, W=96, H=80, seed=21)
    a = {k: np64(v) for k, v in scene.items()}
    s32 = RasterOracle("f32").forward(st, a["means3D"], a["opacities"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
    s64 = RasterOracle("f64").forward(st, a["means3D"], a["opacities"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
    clean = s32["ambiguous"] == 0
    assert clean.mean() > 0.99
    assert np.abs(s32["out_color"] - s64["out_color"])[:, clean].max() < 1e-4
    # empty input: all pixels = background, tidx = -1
    e = RasterOracle("f32").forward(st, np.zeros((0, 3)), np.zeros((0, 1)), shs=np.zeros((0, 16, 3)), scales=np.zeros((0, 3)), rotations=np.zeros((0, 4)))
    assert e["R"] == 0 and (e["out_tidx"] == -1).all()
    np.testing.assert_allclose(e["out_color"][1], st.bg[1], rtol=1e-6)
    # everything behind the camera: culled
    b = a["means3D"].copy()
    b[:, :] = np.asarray(st.campos)[None] * 3
    c = RasterOracle("f32").forward(st, b, a["opacities"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
    assert (c["radii"] == 0).all() and c["R"] == 0
