"""-m gpu: BASELINE.json's configurations at their stated sizes (single-GPU forms).

  c1  10 k Gaussians 400x400 static           forward + backward parity   (tests/test_gpu_raster.py, case config1_like)
  c2  200 k + deformable_field, 800x800       stage-1 forward + backward incl. the MLP's gradients vs the composed oracle
  c3  1 M, 1352x1014                          FULL-size rasterizer parity: discrete results bit-exact vs the f32 oracle, RGB
                                              <= 1e-4 on the pixels the oracle calls unambiguous, gradients vs the f64 oracle
  c5  2 M, K = 512, nn = 8, fp16-operand MLP  single-GPU form: stage-3 render and stage-1 deformation against the fp32 MLP
                                              at the 16-bit tolerance, + size-independent properties
(c4 / c5's 8-GPU legs need a node: the 2-rank step is exercised by tests/test_gpu_training_api.py.)"""
import math
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gaussianprediction_amd as gpa  # noqa: E402
from gaussianprediction_amd.cameras import orbit_cameras  # noqa: E402
from gaussianprediction_amd.rasterizer import raster_forward_debug  # noqa: E402
from gaussianprediction_amd.scene_synth import SceneSpec, make_gaussians, make_keypoints  # noqa: E402
from golden.make_golden import mlp_state  # noqa: E402
from gpu_util import f32_settings, torch_settings  # noqa: E402
from oracle import deform_oracle as do  # noqa: E402
from oracle.oracle import RasterOracle, RasterSettings  # noqa: E402
from util import rel_l2  # noqa: E402

THREADS = min(32, os.cpu_count() or 1)      # the oracle's OpenMP loops stop scaling long before a 256-thread host


def _settings_of(cam, sh_degree=3, bg=(0.0, 0.0, 0.0)):
    n64 = lambda t: t.detach().cpu().double().numpy()     # noqa: E731
    return f32_settings(RasterSettings(image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
                                       tanfovy=math.tan(cam.FoVy * 0.5), bg=np.asarray(bg, np.float64), scale_modifier=1.0,
                                       viewmatrix=n64(cam.world_view_transform), projmatrix=n64(cam.full_proj_transform),
                                       sh_degree=sh_degree, campos=n64(cam.camera_center)))


def test_c3_full_size_rasterizer_parity_forward_and_backward():
    """configs[2]: 1 M Gaussians, 1352 x 1014, one camera of the bench's arc."""
    N, W, H = 1_000_000, 1352, 1014
    raw = make_gaussians(SceneSpec(n_gaussians=N, extent=(1.5, 1.5, 0.5), scale_lo=0.003, scale_hi=0.012))
    sc = dict(means3D=raw["xyz"], opacities=torch.sigmoid(raw["opacity"]), shs=torch.cat([raw["features_dc"], raw["features_rest"]], 1),
              scales=torch.exp(raw["scaling"]), rotations=torch.nn.functional.normalize(raw["rotation"]))
    a = {k: v.numpy().astype(np.float64) for k, v in sc.items()}
    cam = orbit_cameras(8, 4.0, 2 * math.atan(1 / 1.8), W, H, arc_deg=40.0, elevation_deg=5.0)[3]
    st = _settings_of(cam, bg=(0.1, 0.2, 0.3))
    s = RasterOracle("f32", threads=THREADS).forward(st, a["means3D"], a["opacities"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
    dev = {k: v.cuda() for k, v in sc.items()}
    h = raster_forward_debug(torch_settings(st), dev["means3D"], dev["opacities"], shs=dev["shs"], scales=dev["scales"], rotations=dev["rotations"])
    # discrete results: exact, all of them
    np.testing.assert_array_equal(h["radii"].cpu().numpy(), s["radii"])
    assert h["R"] == s["R"] and s["R"] > 3 * N
    np.testing.assert_array_equal(h["ranges"].cpu().numpy(), s["ranges"])
    np.testing.assert_array_equal(h["point_list"].cpu().numpy().astype(np.uint32), s["point_list"][:s["R"]])
    clean = s["ambiguous"] == 0
    assert clean.mean() > 0.99, f"ambiguous fraction {1 - clean.mean():.4f}"
    err = np.abs(h["color"].cpu().numpy().astype(np.float64) - s["out_color"])
    assert err[:, clean].max() <= 1e-4, f"RGB Linf {err[:, clean].max():.3e}"
    assert np.abs(h["depth"][0].cpu().numpy() - s["out_depth"])[clean].max() <= 1e-4 * max(1.0, float(s["out_depth"].max()))
    assert (h["tidx"].cpu().numpy()[clean] == s["out_tidx"][clean]).all()
    nc_equal = (h["n_contrib"].cpu().numpy() == s["n_contrib"])[clean].mean()
    assert nc_equal == 1.0, f"n_contrib differs on {1 - nc_equal:.2e} of the unambiguous pixels"
    print(f"[c3 full size] R={s['R']} ambiguous={1 - clean.mean():.4%} RGB Linf(clean)={err[:, clean].max():.2e} "
          f"RGB Linf(all)={err.max():.2e}")
    # ---- gradients.  At 1352 x 1014 float32 itself limits the agreement with float64: a pixel coordinate near 1000 has a float32
    # spacing of 6e-5 px, so alpha carries ~1e-4 relative rounding and the float32 ORACLE's gradients already sit 1.5e-4 .. 3.3e-4
    # (rel-L2) from the float64 shadow.  Two bars therefore: (1) HIP vs the float32 oracle's backward (same precision class,
    # different formulation: back-to-front recursion, double accumulators) <= 1e-4; (2) HIP is no farther from the float64 shadow
    # than that float32 restatement is.  The shadow shares the float32 forward's visibility and per-tile depth order (depths
    # that coincide in float32 only would otherwise be blended in the opposite order, see forward_with_binning_of).
    rng = np.random.default_rng(3)
    wimg = rng.normal(size=(3, H, W)).astype(np.float32)
    o32, o64 = RasterOracle("f32", threads=THREADS), RasterOracle("f64", threads=THREADS)
    g32 = o32.backward(s, wimg.astype(np.float64))
    s64 = o64.forward_with_binning_of(s, st, a["means3D"], a["opacities"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
    d64 = np.abs(s64["out_color"] - s["out_color"]).max(axis=0)                   # float32 vs float64 image, same order:
    assert float(np.median(d64)) < 2e-6 and float((d64[clean] > 1e-4).mean()) < 1e-3   # 1/255 decisions flip on a few pixels
    # PSNR stand-in for north_star's "within 0.05 dB" (no dataset, no reference rasterizer: SURVEY 8d "otherwise report PSNR of GPU
    # render vs oracle render"): (1) PSNR of the HIP image against the float64 oracle's image of the same Gaussians; (2) both
    # images scored against one pseudo ground truth (the float64 image + 0.05 noise, ~26 dB): the score the evaluation would print
    # [REF utils/image_utils.py:18-20, train.py:267] moves by less than 1e-3 dB
    from host_checkers import psnr
    himg = torch.tensor(h["color"].cpu().numpy().astype(np.float64))
    oimg = torch.tensor(s64["out_color"])
    gt = (oimg + 0.05 * torch.tensor(rng.normal(size=oimg.shape))).clamp(0, 1)
    p_direct, p_hip, p_orc = psnr(himg, oimg), psnr(himg, gt), psnr(oimg, gt)
    print(f"[c3 full size] PSNR(HIP, f64 oracle) = {p_direct:.1f} dB; against a common target: HIP {p_hip:.5f} dB, f64 oracle {p_orc:.5f} dB")
    assert p_direct > 80.0 and abs(p_hip - p_orc) < 1e-3
    g64 = o64.backward(s64, wimg.astype(np.float64))
    L = {k: v.clone().requires_grad_(True) for k, v in dev.items()}
    m2 = torch.zeros(N, 3, device="cuda", requires_grad=True)
    img, radii, depth, tidx = gpa.GaussianRasterizer(raster_settings=torch_settings(st))(
        means3D=L["means3D"], means2D=m2, opacities=L["opacities"], shs=L["shs"], scales=L["scales"], rotations=L["rotations"])
    (img * torch.tensor(wimg, device="cuda")).sum().backward()
    hip = {k: L[k].grad.cpu().numpy() for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    hip["means2D"] = m2.grad[:, :2].cpu().numpy()
    e32 = {k: rel_l2(hip[k], g32[k]) for k in hip}
    e64 = {k: rel_l2(hip[k], g64[k]) for k in hip}
    floor = {k: rel_l2(g32[k], g64[k]) for k in hip}
    print("[c3 full size] gradient rel-L2  HIP vs f32 oracle:", {k: f"{v:.1e}" for k, v in e32.items()})
    print("[c3 full size] gradient rel-L2  HIP vs f64 shadow:", {k: f"{v:.1e}" for k, v in e64.items()},
          " f32 oracle vs f64 shadow:", {k: f"{v:.1e}" for k, v in floor.items()})
    for k in hip:
        assert e32[k] < 1e-4, f"{k}: rel L2 vs the float32 oracle {e32[k]:.3e}"
        assert e64[k] < 1.5 * floor[k] + 2e-5, f"{k}: {e64[k]:.3e} from the float64 shadow, the float32 oracle {floor[k]:.3e}"


def test_large_R_binning_and_composite_parity():
    """SURVEY 7 hard part "stable sort at R ~ 1e7 - 1e8": the c3 scene with 2.4 x larger splats -> R = 12.1 M tile-splat
    instances (R / N = 12): the tile sort runs its > 8 M-key configuration (16 keys per thread, csrc/sort_scan.hip), tile lists
    reach 4 000 entries.  Exact mode and capacity mode against the float32 oracle: every discrete result bit-exact, RGB <= 1e-4
    on the unambiguous pixels; gradients vs the float32 oracle's backward <= 1e-4 (as at c3)."""
    N, W, H = 1_000_000, 1352, 1014
    raw = make_gaussians(SceneSpec(n_gaussians=N, extent=(1.5, 1.5, 0.5), scale_lo=0.007, scale_hi=0.03))
    sc = dict(means3D=raw["xyz"], opacities=torch.sigmoid(raw["opacity"]), shs=torch.cat([raw["features_dc"], raw["features_rest"]], 1),
              scales=torch.exp(raw["scaling"]), rotations=torch.nn.functional.normalize(raw["rotation"]))
    a = {k: v.numpy().astype(np.float64) for k, v in sc.items()}
    cam = orbit_cameras(8, 4.0, 2 * math.atan(1 / 1.8), W, H, arc_deg=40.0, elevation_deg=5.0)[3]
    st = _settings_of(cam, bg=(0.1, 0.2, 0.3))
    o32 = RasterOracle("f32", threads=THREADS)
    s = o32.forward(st, a["means3D"], a["opacities"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
    assert s["R"] >= 12_000_000 and s["R"] > (8 << 20)              # the sort's large-input configuration
    dev = {k: v.cuda() for k, v in sc.items()}
    rs = torch_settings(st)
    h = raster_forward_debug(rs, dev["means3D"], dev["opacities"], shs=dev["shs"], scales=dev["scales"], rotations=dev["rotations"])
    np.testing.assert_array_equal(h["radii"].cpu().numpy(), s["radii"])
    assert h["R"] == s["R"]
    np.testing.assert_array_equal(h["ranges"].cpu().numpy(), s["ranges"])
    np.testing.assert_array_equal(h["point_list"].cpu().numpy().astype(np.uint32), s["point_list"][:s["R"]])
    clean = s["ambiguous"] == 0
    assert clean.mean() > 0.99
    err = np.abs(h["color"].cpu().numpy().astype(np.float64) - s["out_color"])
    assert err[:, clean].max() <= 1e-4, f"RGB Linf {err[:, clean].max():.3e}"
    assert (h["tidx"].cpu().numpy()[clean] == s["out_tidx"][clean]).all()
    assert (h["n_contrib"].cpu().numpy() == s["n_contrib"])[clean].all()
    # capacity mode (no host read of R; sentinel-padded sort over the capacity): bit-identical image and lists
    status = torch.zeros(2, dtype=torch.int32, device="cuda")
    rs_cap = rs._replace(binning_capacity=int(1.05 * s["R"]), binning_status=status)
    hc = raster_forward_debug(rs_cap, dev["means3D"], dev["opacities"], shs=dev["shs"], scales=dev["scales"], rotations=dev["rotations"])
    assert status.tolist() == [s["R"], 0]
    assert torch.equal(hc["color"], h["color"]) and torch.equal(hc["tidx"], h["tidx"]) and torch.equal(hc["ranges"], h["ranges"])
    assert torch.equal(hc["point_list"][:s["R"]], h["point_list"][:s["R"]])
    print(f"[large R] R={s['R']} (R/N = {s['R'] / N:.1f}) longest tile list {int((s['ranges'][:, 1] - s['ranges'][:, 0]).max())} "
          f"ambiguous={1 - clean.mean():.4%} RGB Linf(clean)={err[:, clean].max():.2e}")
    # gradients against the float32 oracle's backward (different formulation, double accumulators)
    wimg = np.random.default_rng(3).normal(size=(3, H, W)).astype(np.float32)
    g32 = o32.backward(s, wimg.astype(np.float64))
    L = {k: v.clone().requires_grad_(True) for k, v in dev.items()}
    m2 = torch.zeros(N, 3, device="cuda", requires_grad=True)
    img, radii, depth, tidx = gpa.GaussianRasterizer(raster_settings=rs)(
        means3D=L["means3D"], means2D=m2, opacities=L["opacities"], shs=L["shs"], scales=L["scales"], rotations=L["rotations"])
    (img * torch.tensor(wimg, device="cuda")).sum().backward()
    hip = {k: L[k].grad.cpu().numpy() for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    hip["means2D"] = m2.grad[:, :2].cpu().numpy()
    e32 = {k: rel_l2(hip[k], g32[k]) for k in hip}
    print("[large R] gradient rel-L2  HIP vs f32 oracle:", {k: f"{v:.1e}" for k, v in e32.items()})
    for k in hip:
        assert e32[k] < 1e-4, f"{k}: rel L2 vs the float32 oracle {e32[k]:.3e}"


def _c2_build(N=200_000, W=800, H=800, it=20000):
    args = SimpleNamespace(beta=0.1, d=4, w=256, feature_dim=32, second_stage_iteration=30000, third_stage_iteration=40000,
                           jointly_iteration=1000, nearest_num=6, norm_rotation=True, step_opacity=False, step_opacity_iteration=5000,
                           opacity_type="implicit", xyz_noise_iteration=0, xyz_freq=10, time_freq=6)
    raw = make_gaussians(SceneSpec(n_gaussians=N, extent=(1.3, 1.3, 1.3), scale_lo=0.005, scale_hi=0.02, seed=2024))
    raw["motion_feature"] = raw["motion_feature"] * 50            # a visibly non-trivial deformation
    sd = {k: torch.tensor(v) for k, v in mlp_state(77, 32 + 60 + 12, 7).items()}
    pc = gpa.GaussianModel(3, args)
    pc.set_inputDim(12, 60)
    d = lambda t: t.cuda()                                        # noqa: E731
    pc.create_from_tensors(d(raw["xyz"]), d(raw["features_dc"]), d(raw["features_rest"]), d(raw["scaling"]), d(raw["rotation"]),
                           d(raw["opacity"]), d(raw["motion_feature"]))
    pc.df_model.load_state_dict(sd)
    cam = orbit_cameras(8, 4.0, 0.6911, W, H, arc_deg=360.0, elevation_deg=20.0, device="cuda")[2]
    P = dict(xyz=raw["xyz"], rotation=raw["rotation"], scaling=raw["scaling"], opacity=raw["opacity"], motion_feature=raw["motion_feature"])
    return pc, cam, P, sd, raw, args


def test_c2_stage1_forward_and_backward_vs_composed_oracle():
    """configs[1]: 200 k Gaussians + deformable_field (MLP over every Gaussian, F = 6), 800 x 800, iteration 20000:
    render forward against deform oracle -> C raster oracle, and d(sum(image * G)) / d(every parameter) against the same chain
    in float64 (torch autograd through the deformation oracle, the C oracle's backward for the rasterizer)."""
    it = 20000
    pc, cam, P, sd, raw, args = _c2_build(it=it)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    time = torch.tensor([0.6], device="cuda")
    pkg = gpa.render(cam, pc, pipe, torch.zeros(3, device="cuda"), time=time, it=it)
    st = _settings_of(cam)
    shs = torch.cat([raw["features_dc"], raw["features_rest"]], 1)
    n64 = lambda t: t.detach().float().numpy().astype(np.float64)         # noqa: E731
    with torch.no_grad():
        xyz, q, s, o = do.deform_forward(P, sd, torch.tensor(0.6), it, args)
        xh, qh, sh_, oh = pc(time, it)            # the very values render() fed the rasterizer (the kernels are deterministic)
    # (A) deformation: HIP (split-fp16 MFMA MLP at fp32 accuracy) vs the torch restatement
    for got, want, tol in ((xh, xyz, 2e-5), (qh, q, 2e-5), (sh_, s, 1e-6), (oh, o, 5e-6)):
        d = float((got.cpu() - want).abs().max())
        assert d <= tol * max(1.0, float(want.abs().max())), f"deformation output off by {d:.2e}"
    # (B) rasterizer on exactly those values: bit-exact discrete results, every unambiguous pixel within 1e-4
    o32 = RasterOracle("f32", threads=THREADS)
    ref = o32.forward(st, n64(xh.cpu()), n64(oh.cpu()), shs=n64(shs), scales=n64(sh_.cpu()), rotations=n64(qh.cpu()))
    img = pkg["render"].detach().cpu().numpy()
    err = np.abs(img - ref["out_color"]).max(axis=0)
    clean = ref["ambiguous"] == 0
    assert clean.mean() > 0.99 and err[clean].max() <= 1e-4, f"RGB Linf {err[clean].max():.2e} on the unambiguous pixels"
    np.testing.assert_array_equal(pkg["radii"].cpu().numpy(), ref["radii"])
    vis = pkg["visibility_filter"].cpu().numpy()
    assert (vis == (ref["radii"] > 0)).all() and vis.sum() > 100_000
    # ---- backward: the whole chain in float64 (torch autograd through the deformation oracle, the C oracle's backward for the
    # rasterizer).  The float64 rasterizer ADOPTS the float32 forward's visibility and per-tile order (forward_with_binning_of):
    # what is compared is the calculus on one and the same set of (pixel, splat) pairs, not two different images
    G = torch.randn(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(5))
    (pkg["render"] * G.cuda()).sum().backward()
    P64 = {k: v.double().requires_grad_(True) for k, v in P.items()}
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    shs64 = shs.double().requires_grad_(True)
    xo, qo, so, oo = do.deform_forward(P64, sd64, torch.tensor(0.6, dtype=torch.float64), it, args)
    orc = RasterOracle("f64", threads=THREADS)
    d64 = lambda t: t.detach().numpy()                                     # noqa: E731
    s64 = orc.forward_with_binning_of(ref, st, d64(xo), d64(oo), shs=d64(shs64), scales=d64(so), rotations=d64(qo))
    g = orc.backward(s64, G.double().numpy())
    g32 = o32.backward(ref, G.double().numpy())         # the float32 oracle's backward: same precision class as the kernels

    def chain(gr):
        """Rasterizer gradients `gr` pushed through the float64 deformation: {parameter: gradient}."""
        for t in list(P64.values()) + list(sd64.values()):
            t.grad = None
        torch.autograd.backward([xo, qo, so, oo], [torch.tensor(gr["means3D"]), torch.tensor(gr["rotations"]), torch.tensor(gr["scales"]),
                                                   torch.tensor(gr["opacities"]).reshape(oo.shape)], retain_graph=True)
        out = {"xyz": P64["xyz"].grad, "rotation": P64["rotation"].grad, "scaling": P64["scaling"].grad, "opacity": P64["opacity"].grad,
               "motion_feature": P64["motion_feature"].grad}
        out = {k: v.numpy().copy() for k, v in out.items()}
        out["features_dc"], out["features_rest"] = gr["shs"][:, :1], gr["shs"][:, 1:]
        for k in sd64:
            out["mlp." + k] = sd64[k].grad.numpy().copy()
        return out

    c64, c32 = chain(g), chain(g32)
    hip = {"xyz": pc._xyz.grad, "rotation": pc._rotation.grad, "scaling": pc._scaling.grad, "opacity": pc._opacity.grad,
           "motion_feature": pc.motion_feature.grad, "features_dc": pc._features_dc.grad, "features_rest": pc._features_rest.grad}
    hip.update({"mlp." + k: p.grad for k, p in pc.df_model.named_parameters()})
    hip = {k: v.cpu().numpy() for k, v in hip.items()}
    e32 = {k: rel_l2(hip[k], c32[k]) for k in hip}
    e64 = {k: rel_l2(hip[k], c64[k]) for k in hip}
    floor = {k: rel_l2(c32[k], c64[k]) for k in hip}
    print("[c2] gradient rel-L2, HIP vs (f32 raster oracle -> f64 deformation):", {k: f"{v:.1e}" for k, v in e32.items()})
    print("[c2] gradient rel-L2, HIP vs the all-f64 chain:", {k: f"{v:.1e}" for k, v in e64.items()}, " f32 raster oracle vs all-f64:",
          {k: f"{v:.1e}" for k, v in floor.items()})
    # two bars, as at c3: (1) against the float32 rasterizer oracle (different formulation, double accumulators) + float64 deformation:
    # SURVEY 8d's 1e-4; (2) no farther from the all-float64 chain than that float32 restatement is (at 800 x 800 float32 pixel
    # coordinates alone put ~3e-4 between ANY float32 rasterizer backward and float64)
    for k in hip:
        assert e32[k] < 1e-4, f"{k}: rel L2 {e32[k]:.3e} vs the float32 rasterizer oracle chain"
        assert e64[k] < 1.5 * floor[k] + 2e-5, f"{k}: {e64[k]:.3e} from the all-float64 chain, the float32 oracle chain {floor[k]:.3e}"


def _c5_build(N, K=512, nn=8, W=800, H=800):
    args = SimpleNamespace(beta=0.1, d=4, w=256, feature_dim=32, second_stage_iteration=30000, third_stage_iteration=40000,
                           jointly_iteration=1000, nearest_num=nn, norm_rotation=True, step_opacity=False, step_opacity_iteration=5000,
                           opacity_type="implicit", xyz_noise_iteration=0)
    raw = make_gaussians(SceneSpec(n_gaussians=N, extent=(1.3, 1.3, 1.3), scale_lo=0.003, scale_hi=0.010, seed=2024), device="cuda")
    kp, kpf, idx, rw = make_keypoints(raw["xyz"], raw["motion_feature"], K, nn)
    torch.manual_seed(2024)
    pc = gpa.GaussianModel(3, args)
    pc.set_inputDim(12, 60)
    pc.create_from_tensors(raw["xyz"], raw["features_dc"], raw["features_rest"], raw["scaling"], raw["rotation"], raw["opacity"],
                           raw["motion_feature"] * 50, kp, kpf * 50)
    pc.set_keypoint_weights(rw, idx)
    cams = orbit_cameras(8, 4.0, 0.6911, W, H, arc_deg=360.0, elevation_deg=20.0, device="cuda")
    return pc, cams


def test_c5_single_gpu_form_fp16_operand_mlp():
    """configs[4] on one GPU: 2 M Gaussians, max_keypoints 512, nearest_num 8, deformation MLP with fp16 MFMA operands."""
    N = 2_000_000
    pc, cams = _c5_build(N)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    t = torch.tensor([0.35], device="cuda")
    bg0, bg1 = torch.zeros(3, device="cuda"), torch.tensor([0.3, 0.6, 0.9], device="cuda")
    with torch.no_grad():
        # stage 1: the MLP over all 2 M rows is the MFMA stress; 16-bit operands vs the fp32 MLP at the 16-bit tolerance
        pc.df_model.precision = "fp32"
        x32, q32, _, _ = pc(t, 20000)
        pc.df_model.precision = "fp16"
        x16, q16, _, _ = pc(t, 20000)
        d32 = x32 - pc._xyz
        scale = float(d32.abs().max())
        assert scale > 1e-3
        assert float((x16 - x32).abs().max()) < 3e-3 * max(scale, 1.0), float((x16 - x32).abs().max())
        assert float((q16 - q32).abs().max()) < 3e-3
        # stage 3 (K = 512, nn = 8): rendered image with the 16-bit MLP vs the fp32 one
        pc.df_model.precision = "fp32"
        r32 = gpa.render(cams[1], pc, pipe, bg0, time=t, it=50000)
        pc.df_model.precision = "fp16"
        r16 = gpa.render(cams[1], pc, pipe, bg0, time=t, it=50000)
        r16b = gpa.render(cams[1], pc, pipe, bg1, time=t, it=50000)
        err = (r16["render"] - r32["render"]).abs()
        assert float(err.median()) < 1e-4 and float((err > 2e-2).float().mean()) < 1e-3, (float(err.median()), float(err.max()))
        assert (r16["radii"] != r32["radii"]).float().mean() < 1e-2
        assert int((r16["radii"] > 0).sum()) > N // 2
        # size-independent properties of the rasterizer at this size: linear in the background with T_final in [0, 1]
        diff = r16b["render"] - r16["render"]
        Tf = diff[0] / 0.3
        assert float(Tf.min()) >= -1e-6 and float(Tf.max()) <= 1 + 1e-6
        assert float((diff[1] - Tf * 0.6).abs().max()) < 1e-5 and float((diff[2] - Tf * 0.9).abs().max()) < 1e-5
        assert torch.equal(r16["tidx"], r16b["tidx"])
    # one full train step of each form runs and leaves finite parameters
    from gaussianprediction_amd.train_step import TrainStep
    gts = [r32["render"].clamp(0, 1)] * len(cams)
    for it in (20000, 50000):
        ts = TrainStep(pc, cams, gts, it)
        loss, pkg = ts.step(1)
        assert math.isfinite(float(loss))
        assert torch.isfinite(pc._xyz).all() and all(torch.isfinite(p).all() for p in pc.df_model.parameters())


def test_c4_single_gpu_form_lifecycle_opacity_batch_of_views():
    """configs[3] on one GPU: 1 M Gaussians, K = 300 keypoints, F = 10, lifecycle opacity (a second MLP pass over all N with its own
    gradient into `_xyz`), several views per iteration accumulated as `--batch` does [REF train.py:113-133, scripts/train/hyper/lemon.sh]
    -- the 8-GPU leg runs the same step with one view per rank and the exchange of tests/test_gpu_training_api.py."""
    from gaussianprediction_amd.train_step import TrainStep
    N, W, H = 1_000_000, 1352, 1014
    args = SimpleNamespace(beta=0.1, d=4, w=256, feature_dim=32, second_stage_iteration=30000, third_stage_iteration=40000,
                           jointly_iteration=1000, nearest_num=6, norm_rotation=True, step_opacity=True, step_opacity_iteration=5000,
                           opacity_type="implicit", xyz_noise_iteration=0)
    raw = make_gaussians(SceneSpec(n_gaussians=N, extent=(1.5, 1.5, 0.5), scale_lo=0.003, scale_hi=0.012, seed=2024), device="cuda")
    kp, kpf, idx, rw = make_keypoints(raw["xyz"], raw["motion_feature"], 300, 6)
    torch.manual_seed(2024)
    pc = gpa.GaussianModel(3, args)
    pc.set_inputDim(20, 60)                                    # F = 10
    pc.create_from_tensors(raw["xyz"], raw["features_dc"], raw["features_rest"], raw["scaling"], raw["rotation"], raw["opacity"],
                           raw["motion_feature"], kp, kpf)
    pc.set_keypoint_weights(rw, idx)
    assert pc.df_model.feature_to_deformation[0].weight.shape[0] == 8 and isinstance(pc.opacity_thres, torch.nn.Parameter)
    cams = orbit_cameras(8, 4.0, 2 * math.atan(1 / 1.8), W, H, arc_deg=40.0, elevation_deg=5.0, device="cuda")
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    with torch.no_grad():
        gts = [gpa.render(c, pc, pipe, torch.zeros(3, device="cuda"), time=torch.tensor([0.3], device="cuda"), it=50000)["render"] * 0.9
               for c in cams[:2]] * 4
    ts = TrainStep(pc, cams, gts, 50000, batch=2)
    names = [g["name"] for g in pc.optimizer.param_groups]
    assert names == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "s_xyz", "s_motion_feature", "df_mlp", "opacity_thres"]
    x0 = pc._xyz.detach().clone()
    loss, pkg = ts.step(0)
    assert math.isfinite(float(loss)) and pc.lifecycle_opacity is not None
    assert pkg["radii"].shape == (N,) and int(pkg["visibility_filter"].sum()) > N // 2
    assert torch.isfinite(pc._xyz).all() and not torch.equal(pc._xyz.detach(), x0)
    assert all(torch.isfinite(p).all() for p in pc.df_model.parameters())


def test_c3_fused_step_equals_the_graph_step_at_full_size():
    """configs[2] at FULL size (1 M Gaussians, 1352 x 1014, K = 250) -- the bench's own workload: three steps at zero learning rates
    through gp_train_step_run and through the autograd graph leave the same Adam moments (= the same gradients, every tensor) and
    the same losses; the step the bench times is the step the parity tests check."""
    import bench
    from gaussianprediction_amd.train_step import TrainStep
    args = SimpleNamespace(gaussians=1_000_000, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000,
                           scale_lo=0.003, scale_hi=0.012)
    zero = dict(xyz=0.0, f_dc=0.0, opacity=0.0, scaling=0.0, rotation=0.0, kpts=0.0, mlp=0.0)
    res = []
    for fused in (True, False):
        pc, cams, gts, margs = bench.build_workload(args, torch.device("cuda", 0))
        ts = TrainStep(pc, cams, gts, 50000, lrs=zero, speculative=True, fused=fused)
        pre = len(cams) + TrainStep.SPEC_SLOTS
        losses = [float(ts.step(i)[0]) for i in range(pre + 3)]
        torch.cuda.synchronize()
        assert ts.fused_steps == (3 if fused else 0) and ts.redone == 0
        sd = pc.optimizer.state_dict()
        res.append(dict(loss=losses, state={k: v["exp_avg"].clone() for k, v in sd["state"].items()}, names=[g["name"] for g in sd["param_groups"]]))
        del pc, ts
        torch.cuda.empty_cache()
    a, b = res
    assert np.allclose(a["loss"], b["loss"], rtol=5e-6, atol=1e-7), (a["loss"], b["loss"])
    for k in a["state"]:
        e = float((a["state"][k] - b["state"][k]).norm() / b["state"][k].norm().clamp_min(1e-30))
        assert e < 5e-5, (k, e)
