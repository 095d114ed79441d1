"""-m gpu: the HIP training path against an independent differentiable renderer over a whole (small) training run -- the stand-in
for north_star's "PSNR within 0.05 dB of reference on D-NeRF bouncingballs" (the dataset is not available offline and the
reference rasterizer is an empty submodule).

A small dynamic scene (400 Gaussians, 56 x 56, 8 views) is trained for 304 iterations -- stage 1 (per-Gaussian deformation MLP),
then stage 3 (keypoints + blend), each with the optimizer the reference creates for it -- by tests/host_checkers.DenseRefTrainer:
torch restatement of GaussianModel.forward -> tests/dense_ref.py (dense per-pixel renderer, no tiles, no hand-derived gradients)
-> torch L1 + SSIM -> torch.autograd -> torch.optim.Adam(eps=1e-15), all float64.  Against that run:

 (a) TEACHER-FORCED, every 4th iteration: the reference's current parameters and Adam moments are loaded into a HIP model, ONE
     step of this package's TrainStep is taken (render -> fused L1+SSIM -> HIP backward -> fused Adam, float32), and its parameter
     UPDATE is compared with the reference's own update at that iteration, tensor by tensor.  76 states along a real trajectory,
     early and late, both stages: a statement about the implementation that chaos cannot blur.
 (b) FREE-RUNNING: a second HIP model trains on its own from the same start.  3DGS training is a chaotic map (depth-order
     swaps and the 1/255 / 1e-4 decisions make the loss discontinuous in the parameters, Adam with eps = 1e-15 moves every
     parameter by ~lr per step whatever the gradient's size), so float32 and float64 trajectories separate after ~50 steps
     by amounts that have nothing to do with correctness; the run is held to a stated band instead: both reach the same PSNR
     (mean over the last five passes, +- 1 dB: a control run of the SAME float32 implementation from a start perturbed by
     1e-7 drifts by as much) and the per-pass loss curves stay within 10 % on average.
[REF train.py:101-133,196-197,258-282; utils/image_utils.py:18-20]"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gaussianprediction_amd as gpa  # noqa: E402
from gaussianprediction_amd.cameras import orbit_cameras  # noqa: E402
from gaussianprediction_amd.scene_synth import SceneSpec, make_gaussians, make_keypoints  # noqa: E402
from gaussianprediction_amd.train_step import TrainStep  # noqa: E402
from golden.make_golden import mlp_state  # noqa: E402
from host_checkers import DenseRefTrainer, psnr  # noqa: E402

N, K, W, H, VIEWS = 400, 24, 56, 56, 8
STEPS_STAGE1, STEPS_STAGE3 = 152, 152          # 19 passes over the eight views each
IT1, IT3 = 20000, 50000


def _args():
    return SimpleNamespace(beta=0.1, d=4, w=256, feature_dim=32, second_stage_iteration=30000, third_stage_iteration=40000,
                           jointly_iteration=1000, nearest_num=6, norm_rotation=True, step_opacity=False, step_opacity_iteration=5000,
                           opacity_type="implicit", xyz_noise_iteration=0, xyz_freq=10, time_freq=6)


def _scene(seed):
    args = _args()
    raw = make_gaussians(SceneSpec(n_gaussians=N, scale_lo=0.04, scale_hi=0.16, seed=seed))
    kp, kpf, idx, raw_w = make_keypoints(raw["xyz"], raw["motion_feature"], K, args.nearest_num)
    raw["motion_feature"] = raw["motion_feature"] * 50           # a visible stage-1 deformation
    kpf = kpf * 50
    sd = {k: torch.tensor(v) for k, v in mlp_state(seed + 70, 32 + 60 + 12, 7).items()}
    P = dict(xyz=raw["xyz"], features_dc=raw["features_dc"], features_rest=raw["features_rest"], rotation=raw["rotation"],
             scaling=raw["scaling"], opacity=raw["opacity"], motion_feature=raw["motion_feature"], super_gaussians=kp,
             super_gaussians_feature=kpf)
    return args, P, sd, raw_w, idx


def _hip_model(args, P, sd, raw_w, idx):
    pc = gpa.GaussianModel(3, args)
    pc.set_inputDim(2 * args.time_freq, 6 * args.xyz_freq)
    d = "cuda"
    pc.create_from_tensors(P["xyz"].to(d), P["features_dc"].to(d), P["features_rest"].to(d), P["scaling"].to(d), P["rotation"].to(d),
                           P["opacity"].to(d), P["motion_feature"].to(d), P["super_gaussians"].to(d), P["super_gaussians_feature"].to(d))
    pc.df_model.load_state_dict(sd)
    pc.set_keypoint_weights(raw_w.to(d), idx.to(d))
    return pc


def _load_state(pc, ref, k_stage):
    """Reference parameters + Adam moments -> the HIP model (float32 copies); the optimizer's step counter follows."""
    mods = dict(pc.df_model.state_dict())
    with torch.no_grad():
        for name, p in ref.named_parameters():
            dst = mods[name[len("df_model."):]] if name.startswith("df_model.") else getattr(pc, name)
            dst.copy_(p.detach().float())
            m, v, _ = ref.adam_state(p)
            owner = next(q for q in pc.bucket.params if q.data_ptr() == dst.data_ptr())
            pc.optimizer.load_full_moments(owner, m.float().cuda(), v.float().cuda())
    pc.optimizer.step_count = k_stage


def _snapshot(pc, ref):
    mods = dict(pc.df_model.state_dict())
    return {name: (mods[name[len("df_model."):]] if name.startswith("df_model.") else getattr(pc, name)).detach().double().cpu().clone()
            for name, _ in ref.named_parameters()}


def test_training_follows_an_independent_differentiable_renderer():
    torch.manual_seed(0)
    torch.set_num_threads(min(16, torch.get_num_threads()))       # the float64 reference's tensors are small: more threads only add overhead
    args, P, sd, raw_w, idx = _scene(11)
    cams = orbit_cameras(VIEWS, 4.0, 0.6911, W, H, device="cuda")
    # targets: a "teacher" -- the same scene with displaced, recoloured Gaussians and its own motion -- rendered by the dense
    # reference at the eight (camera, time) pairs
    g = torch.Generator().manual_seed(5)
    Pt = {k: v.clone() for k, v in P.items()}
    Pt["xyz"] = P["xyz"] + 0.03 * torch.randn(P["xyz"].shape, generator=g)
    Pt["features_dc"] = P["features_dc"] + 0.4 * torch.randn(P["features_dc"].shape, generator=g)
    Pt["opacity"] = P["opacity"] + 0.5 * torch.randn(P["opacity"].shape, generator=g)
    Pt["motion_feature"] = P["motion_feature"] * 1.5
    teacher = DenseRefTrainer(Pt, {k: torch.tensor(v) for k, v in mlp_state(999, 104, 7).items()}, args, cams, [torch.zeros(3, H, W)] * VIEWS,
                              raw_w, idx)
    with torch.no_grad():
        gts64 = [teacher.render(v, IT1).clamp(0, 1) for v in range(VIEWS)]
    gts32 = [t.float().cuda() for t in gts64]

    ref = DenseRefTrainer(P, sd, args, cams, gts64, raw_w, idx)
    pc = _hip_model(args, P, sd, raw_w, idx)            # free-running
    pc_tf = _hip_model(args, P, sd, raw_w, idx)         # teacher-forced
    # control for (b): the same float32 run from a start that differs in the last bit (positions x (1 + 1e-7)): how far two runs of
    # ONE implementation drift apart is the yardstick for how far two implementations may
    P2 = dict(P, xyz=P["xyz"] * (1.0 + 1e-7))
    pc_ctl = _hip_model(args, P2, sd, raw_w, idx)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    bg = torch.zeros(3, device="cuda")
    with torch.no_grad():
        p0_hip = float(np.mean([psnr(gpa.render(cams[v], pc, pipe, bg, time=torch.from_numpy(cams[v].time).float().cuda(), it=IT1)["render"],
                                     gts32[v]) for v in range(VIEWS)]))
        p0_ref = float(np.mean([psnr(ref.render(v, IT1), gts64[v]) for v in range(VIEWS)]))
    assert abs(p0_hip - p0_ref) < 0.01, (p0_hip, p0_ref)                      # same start: the two renderers agree
    loss = {"hip": [], "ref": [], "ctl": []}
    tpsnr = {"hip": [], "ref": [], "ctl": []}
    worst = {}
    step = 0
    for it, n in ((IT1, STEPS_STAGE1), (IT3, STEPS_STAGE3)):
        ts = TrainStep(pc, cams, gts32, it)                                     # (re-)creates the stage's optimizer, as train.py does
        ts_tf = TrainStep(pc_tf, cams, gts32, it)
        ts_ctl = TrainStep(pc_ctl, cams, gts32, it)
        ref.set_stage(it, {g_["name"]: g_["lr"] for g_ in pc.optimizer.param_groups})
        lr_of = {name: g_["lr"] for g_ in ref.opt.param_groups for name, p_ in ref.named_parameters() if any(p_ is q for q in g_["params"])}
        for k in range(n):
            v = step % VIEWS
            forced = k % 4 == 0
            if forced:                                   # (a) one HIP step from the reference's exact state
                _load_state(pc_tf, ref, k)
                before = _snapshot(pc_tf, ref)
                ts_tf.step(v)
                after = _snapshot(pc_tf, ref)
                ref_before = {name: p.detach().clone() for name, p in ref.named_parameters()}
            l_hip, pkg = ts.step(v)                      # (b) free-running
            loss["hip"].append(float(l_hip))
            tpsnr["hip"].append(psnr(pkg["render"].detach(), gts32[v]))
            l_ctl, pkg_ctl = ts_ctl.step(v)
            loss["ctl"].append(float(l_ctl))
            tpsnr["ctl"].append(psnr(pkg_ctl["render"].detach(), gts32[v]))
            loss["ref"].append(ref.step(v))
            tpsnr["ref"].append(psnr(ref.last_image, gts64[v]))
            if forced:
                for name, p in ref.named_parameters():
                    d_ref = p.detach() - ref_before[name]
                    d_hip = after[name] - before[name]
                    # in units of the learning rate (Adam's step is ~lr whatever the gradient's size)
                    err = (d_hip - d_ref).abs() / lr_of[name]
                    # rel-L2 of the update, WITHOUT the 0.1 % of the elements that are furthest off (their number is bounded by the
                    # second figure): one hidden unit of one keypoint whose pre-activation lies within rounding of zero is on in one
                    # summation order and off in another, and moves a few dozen weights of the next layers by up to one lr -- round 6's
                    # feature-split MLP forward (another, measurably MORE accurate summation order: tools/probe/mlp_split_ab.py) met such
                    # a unit at one of the 76 states (30 of 65 536 elements; 4.8e-2 untrimmed), the 16-row kernels had not
                    e_abs = (d_hip - d_ref).abs().flatten()
                    n_trim = int(e_abs.numel() // 1000)
                    e_kept = e_abs if n_trim == 0 else e_abs.sort().values[:e_abs.numel() - n_trim]
                    rel = float(e_kept.norm() / d_ref.norm().clamp_min(1e-300))
                    w = worst.setdefault(name.split(".")[0] if name.startswith("df_model") else name, [0.0, 0.0, 0.0])
                    w[0] = max(w[0], rel); w[1] = max(w[1], float((err > 0.05).double().mean())); w[2] = max(w[2], float(err.median()))
            step += 1
    print("[teacher-forced] worst over 76 states, per tensor: rel-L2 of the update (99.9 % of the elements) | fraction of elements off by > 0.05 lr | median error / lr")
    for name, (rel, frac, med) in worst.items():
        print(f"    {name:26s} {rel:9.2e} {frac:9.2e} {med:9.2e}")
    lh, lr_ = np.array(loss["hip"]), np.array(loss["ref"])
    eh, er = lh.reshape(-1, VIEWS).mean(1), lr_.reshape(-1, VIEWS).mean(1)       # per pass over the eight views
    edev = np.abs(eh - er) / er
    tail = 5 * VIEWS
    p1_hip, p1_ref, p1_ctl = (float(np.mean(tpsnr[k_][-tail:])) for k_ in ("hip", "ref", "ctl"))
    lc = np.array(loss["ctl"])
    cdev = np.abs(lc.reshape(-1, VIEWS).mean(1) - eh) / eh
    print(f"[free-running] PSNR {p0_hip:.3f} / {p0_ref:.3f} dB at the start -> {p1_hip:.3f} (HIP, float32) / {p1_ref:.3f} (dense reference, float64) dB "
          f"(mean of the last five passes); loss {lh[0]:.5f} -> {eh[-1]:.5f} / {er[-1]:.5f}; per-pass loss deviation mean {edev.mean():.2e} "
          f"max {edev.max():.2e}; first 24 steps max {np.abs(lh[:24] - lr_[:24]).max() / lr_[:24].min():.2e}")
    print(f"[free-running] control (the HIP run again from a start perturbed by 1e-7): {p1_ctl:.3f} dB, per-pass loss deviation from the first "
          f"HIP run mean {cdev.mean():.2e} max {cdev.max():.2e}")
    for name, (rel, frac, med) in worst.items():
        assert rel <= 2e-2 and frac <= 1e-3 and med <= 1e-3, (name, rel, frac, med)
    assert p1_ref > p0_ref + 5.0 and p1_hip > p0_hip + 5.0                      # both actually train
    assert abs(p1_hip - p1_ref) <= 1.0, f"PSNR (last five passes) differs by {abs(p1_hip - p1_ref):.3f} dB"
    assert edev.mean() <= 0.10 and np.abs(lh[:24] - lr_[:24]).max() <= 5e-3 * lr_[:24].min()   # together until chaos sets in
