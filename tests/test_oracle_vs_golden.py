"""CPU: every oracle piece and host-side helper against golden vectors produced by RUNNING the
reference's importable Python (tests/golden/make_golden.py)."""
import math
import os

import numpy as np
import torch

from golden.make_golden import mlp_state
from gaussianprediction_amd import cameras
from oracle import deform_oracle as do
from oracle.oracle import RasterOracle, RasterSettings

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def test_positional_encoding():
    g = load("posenc.npz")
    out = do.positional_encoding(torch.tensor(g["xyz"]), 10).numpy()
    np.testing.assert_allclose(out, g["xyz_pe10"], rtol=0, atol=2e-6)
    for F in (6, 8, 10):
        for t in (0.0, 0.3, 1.0):
            out = do.positional_encoding(torch.tensor([t], dtype=torch.float32), F).numpy()
            np.testing.assert_allclose(out, g[f"t{t}_F{F}"], rtol=0, atol=2e-6)
    # t=0.3, F=6 -> [0.2955, 0.9553, 0.5646, 0.8253, ...]
    np.testing.assert_allclose(g["t0.3_F6"][:4], [0.2955, 0.9553, 0.5646, 0.8253], atol=1e-4)


def test_deformable_field_forward_backward():
    g = load("deformable_field.npz")
    for tag in ("a", "b", "c"):
        d_in, d_out, M, seed = [int(v) for v in g[f"{tag}_meta"]]
        sd = {k: torch.tensor(v, requires_grad=True) for k, v in mlp_state(seed, d_in, d_out).items()}
        x = torch.tensor(np.random.default_rng(seed + 100).uniform(-1, 1, size=(M, d_in)).astype(np.float32), requires_grad=True)
        gy = torch.tensor(np.random.default_rng(seed + 200).normal(size=(M, d_out)).astype(np.float32))
        y = do.mlp_forward(sd, x)
        (y * gy).sum().backward()
        np.testing.assert_allclose(y.detach().numpy(), g[f"{tag}_y"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(x.grad.numpy(), g[f"{tag}_dx"], rtol=1e-4, atol=1e-6)
        for k, p in sd.items():
            if tag == "a":
                np.testing.assert_allclose(p.grad.numpy(), g[f"a_grad_{k}"], rtol=1e-4, atol=1e-5)
            else:
                gs = g[f"{tag}_gradsum_{k}"]
                assert abs(p.grad.double().sum().item() - gs[0]) <= 1e-3 * max(1.0, gs[1])


def test_quat_mul():
    g = load("quat_mul.npz")
    out = do.quat_mul(torch.tensor(g["q1"]), torch.tensor(g["q2"])).numpy()
    np.testing.assert_allclose(out, g["q1q2"], rtol=1e-6, atol=1e-6)
    i, j = torch.tensor([[0.0, 1, 0, 0]]), torch.tensor([[0.0, 0, 1, 0]])
    np.testing.assert_allclose(do.quat_mul(i, j).numpy(), [[0, 0, 0, 1]])  # i (x) j = k


def test_camera_matrices():
    g = load("cameras.npz")
    for i in range(6):
        cam = cameras.Camera(R=g[f"{i}_R"], T=g[f"{i}_T"], FoVx=float(g[f"{i}_fov"][0]), FoVy=float(g[f"{i}_fov"][1]),
                             width=64, height=48, trans=tuple(g[f"{i}_trans"]), scale=float(g[f"{i}_scale"]))
        np.testing.assert_allclose(cam.world_view_transform.numpy(), g[f"{i}_view"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(cam.projection_matrix.numpy(), g[f"{i}_proj"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(cam.full_proj_transform.numpy(), g[f"{i}_full"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(cam.camera_center.numpy(), g[f"{i}_center"], rtol=1e-5, atol=1e-5)


def _identity_settings(campos, W=64, H=64, sh_degree=3, look_neg_z=False):
    # camera at `campos` looking down +z (or -z): p_view = Rv (p - campos)
    V = np.eye(4)
    if look_neg_z:
        V[1, 1] = V[2, 2] = -1.0
    V[:3, 3] = -V[:3, :3] @ campos
    P = cameras.projection_matrix(0.01, 100.0, 1.0, 1.0).numpy().astype(np.float64)
    view_t = V.T
    full_t = view_t @ P.T
    return RasterSettings(image_height=H, image_width=W, tanfovx=math.tan(0.5), tanfovy=math.tan(0.5),
                          bg=np.zeros(3), scale_modifier=1.0, viewmatrix=view_t, projmatrix=full_t, sh_degree=sh_degree,
                          campos=campos.astype(np.float64))


def test_oracle_sh_matches_reference_eval_sh():
    g = load("sh.npz")
    campos = g["campos"].astype(np.float64)
    dirs = (g["points"] - g["campos"][None]).astype(np.float64)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    p = campos[None] + 2.0 * dirs                    # same directions from the camera as the golden vectors
    n = len(p)
    for deg in range(4):
        st = _identity_settings(campos, sh_degree=deg, look_neg_z=True)
        for prec in ("f32", "f64"):
            s = RasterOracle(prec).preprocess(st, p, np.full((n, 1), 0.5), shs=g["features"], scales=np.full((n, 3), 0.05),
                                              rotations=np.tile([1.0, 0, 0, 0], (n, 1)))
            vis = s["radii"] > 0
            assert vis.sum() > 40
            np.testing.assert_allclose(s["rgb"][vis], g[f"rgb_deg{deg}"][vis], rtol=0, atol=1e-5)


def test_oracle_cov3d_matches_reference():
    g = load("cov3d.npz")
    campos = np.zeros(3)
    st = _identity_settings(campos)
    n = len(g["scales"])
    pts = np.tile([0.0, 0.0, 3.0], (n, 1))
    # the reference's build_rotation normalises q; the rasterizer receives normalised q (get_rotation_)
    q = g["quats"] / np.linalg.norm(g["quats"], axis=1, keepdims=True)
    for prec in ("f32", "f64"):
        s = RasterOracle(prec).preprocess(st, pts, np.full((n, 1), 0.5), shs=np.zeros((n, 16, 3)), scales=g["scales"], rotations=q)
        assert (s["radii"] > 0).all()
        np.testing.assert_allclose(s["cov3D"], g["cov3D"], rtol=2e-5, atol=1e-9)


def test_loss_matches_reference():
    g = load("loss.npz")
    img = torch.tensor(g["img"], requires_grad=True)
    gt = torch.tensor(g["gt"])
    l1 = do.l1_loss(img, gt)
    ss = do.ssim(img, gt)
    loss = 0.8 * l1 + 0.2 * (1.0 - ss)
    loss.backward()
    assert abs(l1.item() - float(g["l1"])) < 1e-6
    assert abs(ss.item() - float(g["ssim"])) < 1e-5
    np.testing.assert_allclose(img.grad.numpy(), g["dimg"], rtol=1e-3, atol=1e-8)
