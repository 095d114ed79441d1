"""-m gpu: a whole training schedule, its restart and its evaluation on the drop-in surfaces
(`from gaussian_renderer import render, render_motion, GaussianModel`), driven by this package's own harness
(train_step.TrainStep + densify.py) -- with the stage thresholds compressed so that 90 iterations cross every phase the
reference's schedule has [REF train.py:36-201, eval.py:226-260]:
warm-up (static) -> stage 1 (per-Gaussian MLP; densify / prune / opacity reset) -> the hook at second_stage_iter + 1 (k-means
keypoints, stage-2 optimizer) -> keypoint growth -> the hook at third_stage_iter + 1 -> stage 3 (hash-grid weights model + kNN
evaluated inside forward) -> checkpoint tuple -> a fresh model sized from it (N_pcd_init / final_kpts_num, create_from_pcd,
training_setup, restore, load_state_dict) that continues to train -> eval.py's restore-without-training_setup and its render loop.
Every name the reference's scripts use on these surfaces is pinned separately, as data: tests/golden/api_surface.json
(tests/test_api_surface.py)."""
import os
from random import Random
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gaussian_renderer import GaussianModel, render, render_motion  # noqa: E402  (the import-name shims)
from gaussianprediction_amd.cameras import orbit_cameras  # noqa: E402
from gaussianprediction_amd import densify as dn  # noqa: E402
from gaussianprediction_amd.train_step import TrainStep  # noqa: E402
from gaussianprediction_amd.training import default_training_args  # noqa: E402

W, H = 96, 80


def _args():
    return SimpleNamespace(beta=0.1, d=4, w=256, feature_dim=32, jointly_iteration=10, second_stage_iteration=40, third_stage_iteration=60,
                           nearest_num=6, norm_rotation=True, step_opacity=False, step_opacity_iteration=5000, opacity_type="implicit",
                           xyz_noise_iteration=0, max_points=24, adaptive_points_num=8, adaptive_from_iter=5, adaptive_end_iter=18,
                           adaptive_interval=5, densify_from_grad="True", densify_from_teaching=False, teaching_threshold=0.2,
                           knn_type="hybird", feature_amplify=5.0, max_gaussian_size=3000, time_noise_ratio=0.5, time_noise_iteration=30,
                           use_time_decay=True)


def _opt():
    return default_training_args(iterations=90, densify_from_iter=12, densification_interval=10, opacity_reset_interval=30,
                                 densify_until_iter=38, densify_grad_threshold=1e-6, position_lr_max_steps=90)


def _scene():
    rng = np.random.default_rng(2)
    pts = rng.uniform(-1.0, 1.0, size=(1500, 3)).astype(np.float32)
    cols = rng.uniform(0.1, 0.9, size=(1500, 3)).astype(np.float32)
    cams = orbit_cameras(6, 4.0, 0.69, W, H, device="cuda")
    g = torch.Generator().manual_seed(3)
    for c in cams:                                    # what Camera.original_image holds
        c.original_image = torch.rand(3, H, W, generator=g).cuda() * 0.5 + 0.2
    return SimpleNamespace(points=pts, colors=cols, normals=np.zeros_like(pts)), cams


def _run(model, cams, gts, opt, args, first, last, rnd, log, batch=1):
    """This package's own iteration (train_step.TrainStep: render -> fused L1 + SSIM -> backward -> fused Adam, `batch` views per
    step) driven over iterations first..last, with the loop-side calls of densify.py after every step."""
    ts = TrainStep(model, cams, gts, first, lambda_dssim=opt.lambda_dssim, batch=batch, schedule=True, training_args=opt)
    groups, pending = len(cams) // batch, []
    for it in range(first, last + 1):
        ts.iteration = it
        if it % 20 == 0:
            model.oneupSHdegree()
        if not pending:
            pending = list(range(groups))
        v = pending.pop(rnd.randint(0, len(pending) - 1))
        # time jitter, decaying to zero over time_noise_iteration (restarted, twice as long, when the keypoints take over)
        start = model.second_stage_iter if (args.use_time_decay and it >= model.second_stage_iter) else 0
        span = args.time_noise_iteration * (2 if start else 1)
        jitter = torch.randn(1, device="cuda") * (args.time_noise_ratio / len(cams)) * (1.0 - min(1.0, (it - start) / span))
        # (the groups whose update the reference's loop skips on this iteration: it densifies / prunes before optimizer.step())
        loss, pkg = ts.step(v, time_offset=jitter, hold=dn.held_groups(model, it, opt))
        log["loss"].append(float(loss)), log["n"].append(model.get_xyz.shape[0])
        log["k"].append(model.super_gaussians.shape[0])
        with torch.no_grad():
            if it < opt.densify_until_iter:
                dn.track_view(model, pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"])
                dn.densification_step(model, it, opt, 2.0, max_gaussian_size=args.max_gaussian_size)
            dn.keypoint_growth_step(model, it, opt, args, pkg["visibility_filter"], pkg["radii"], pkg["viewspace_points"])
    ts.sync_params()
    return last


@pytest.mark.parametrize("batch", [1, 2])
def test_full_schedule_restart_and_eval(tmp_path, batch):
    torch.manual_seed(0)
    args, opt = _args(), _opt()
    pcd, cams = _scene()
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    background = torch.tensor([0, 0, 0], dtype=torch.float32, device="cuda")
    gaussians = GaussianModel(3, args)
    gaussians.set_inputDim(2 * 6, 6 * 10)
    gaussians.create_from_pcd(pcd, 2.0)                            # (Scene.__init__)
    gaussians.training_setup(opt)
    gts = [c.original_image for c in cams]
    log = dict(loss=[], n=[], k=[])
    rnd = Random(0)
    last = 75
    _run(gaussians, cams, gts, opt, args, 1, last, rnd, log, batch)
    L = np.array(log["loss"])
    assert np.isfinite(L).all()
    # it trains in every stage: stage 1 up to the opacity reset at 30 (which blacks the image out: the loss jumps), the recovery
    # after it, and stage 3 once the weights model drives the motion (stage 2, 41..60, only moves the keypoints)
    if batch == 1:
        assert L[25:30].mean() < L[0:5].mean() - 0.01 and L[30] > L[29] + 0.1 and L[36:40].mean() < L[30:34].mean() - 0.02
    assert L[70:75].mean() < L[41:46].mean() - 0.04 / batch
    assert gaussians.active_sh_degree == 3
    assert log["n"][0] == 1500 and log["n"][-1] > 1500                                             # densify / prune changed the cloud
    assert gaussians.second_stage and gaussians.third_stage
    assert log["k"][45] == 24 and log["k"][-1] > 24                                                # k-means keypoints, then growth
    assert [g["name"] for g in gaussians.optimizer.param_groups][:3] == ["xyz", "f_dc", "f_rest"]   # stage-3 groups
    assert any(g["name"] == "weight_mlp" for g in gaussians.optimizer.param_groups)
    # ---- checkpoint tuple [REF train.py:199-201] and the restart path [REF train.py:48-57]
    path = os.path.join(tmp_path, "chkpnt.pth")
    torch.save((gaussians.state_dict(), gaussians.optimizer.state_dict(), last), path)
    (model_params, opt_dict, first_iter) = torch.load(path, weights_only=False)
    g2 = GaussianModel(3, args)
    g2.set_inputDim(12, 60)
    g2.N_pcd_init = model_params["_xyz"].shape[0]
    g2.active_sh_degree = g2.max_sh_degree
    g2.final_kpts_num = model_params["super_gaussians"].shape[0] if "super_gaussians" in model_params.keys() else None
    g2.create_from_pcd(pcd, 2.0)
    g2.training_setup(opt)
    g2.restore(opt_dict, opt, first_iter)
    g2.load_state_dict(model_params, strict=False)
    assert g2.active_sh_degree == 3 and g2._xyz.shape == gaussians._xyz.shape and g2.super_gaussians.shape == gaussians.super_gaussians.shape
    for (n1, p1), (n2, p2) in zip(gaussians.named_parameters(), g2.named_parameters()):
        assert n1 == n2 and torch.equal(p1.detach(), p2.detach()), n1
    assert g2.optimizer.state_dict()["state"].keys() == opt_dict["state"].keys()
    time_ = torch.from_numpy(cams[2].time).float().cuda()
    with torch.no_grad():
        a = render(cams[2], gaussians, pipe, background, time=time_, it=last + 1)["render"]
        b = render(cams[2], g2, pipe, background, time=time_, it=last + 1)["render"]
    assert torch.equal(a, b)                                       # the restored model renders the same image, bit for bit
    log2 = dict(loss=[], n=[], k=[])
    _run(g2, cams, gts, opt, args, first_iter + 1, 90, Random(1), log2, batch)
    assert np.isfinite(log2["loss"]).all() and np.mean(log2["loss"][-5:]) < L[70:75].mean() + 0.02
    # ---- eval.py:226-247: restore WITHOUT training_setup, then the render loop and render_motion [REF eval.py:126,153,205-224]
    with torch.no_grad():
        g3 = GaussianModel(3, args)
        g3.set_inputDim(12, 60)
        g3.N_pcd_init, g3.active_sh_degree = model_params["_xyz"].shape[0], g3.max_sh_degree
        g3.final_kpts_num = model_params["super_gaussians"].shape[0]
        g3.create_from_pcd(pcd, 2.0)
        g3.restore(opt_dict, opt, first_iter)
        g3.load_state_dict(model_params, strict=False)
        for view in cams:
            time_ = torch.from_numpy(view.time).to(torch.float32).cuda()
            pkg = render(view, g3, pipe, background, delta=None, time=time_, it=first_iter)
            assert pkg["render"].shape == (3, H, W) and torch.isfinite(pkg["render"]).all() and (pkg["tidx"] >= -1).all()
        xyz_t, r_t, s_t, o_t, wx, wr = g3(time_, first_iter, return_weights=True)
        pkg2 = render_motion(cams[-1], g3, pipe, background, xyz_t=xyz_t, r_t=r_t, opacity=o_t)
        assert torch.allclose(pkg2["render"], pkg["render"], atol=1e-6) and set(pkg2) == {"render", "viewspace_points", "visibility_filter", "radii"}
        assert torch.equal(render(cams[2], g3, pipe, background, time=torch.from_numpy(cams[2].time).float().cuda(), it=last + 1)["render"], a)


# ---- the same compressed schedule, view-parallel: two ranks on one GPU (gloo) --------------------------------------------------
def _schedule_views(rnd, groups, pending):
    if not pending:
        pending.extend(range(groups))
    return pending.pop(rnd.randint(0, len(pending) - 1))


def _run_pairs(model, cams, gts, opt, args, last, world, rank, batch, log, sharded=None):
    """Iterations 1..last; every step covers one PAIR of views (2 g, 2 g + 1): two ranks with one view each, or one process with
    batch 2.  The time jitter comes from a seeded CPU stream so that both forms see the same numbers."""
    ts = TrainStep(model, cams, gts, 1, lambda_dssim=opt.lambda_dssim, batch=batch, schedule=True, training_args=opt, sharded=sharded)
    rnd, pending, jit = Random(0), [], torch.Generator().manual_seed(9)
    for it in range(1, last + 1):
        ts.iteration = it
        g = _schedule_views(rnd, len(cams) // 2, pending)
        start = model.second_stage_iter if it >= model.second_stage_iter else 0
        span = args.time_noise_iteration * (2 if start else 1)
        jitter = (torch.randn(1, generator=jit) * (args.time_noise_ratio / len(cams)) * (1.0 - min(1.0, (it - start) / span))).cuda()
        loss, pkg = ts.step(g * world + rank if world > 1 else g, time_offset=jitter, hold=dn.held_groups(model, it, opt))
        with torch.no_grad():
            if it < opt.densify_until_iter:
                dn.track_view(model, pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"])
                out = dn.densification_step(model, it, opt, 2.0, max_gaussian_size=args.max_gaussian_size)
                log["densify"] += out[0] is not None
                log["prune"] += out[2] is not None
                log["reset"] += it % opt.opacity_reset_interval == 0
            log["grown"] += bool(dn.keypoint_growth_step(model, it, opt, args, pkg["visibility_filter"], pkg["radii"], pkg["viewspace_points"]))
        log["loss"].append(float(loss)), log["n"].append(model.get_xyz.shape[0]), log["k"].append(model.super_gaussians.shape[0])
    ts.sync_params()


def _opt_fast():
    """Densification every 6 iterations from 6 on (three doublings fit under the cap of 7000 before iteration 38), reset at 30."""
    return default_training_args(iterations=90, densify_from_iter=5, densification_interval=6, opacity_reset_interval=30,
                                 densify_until_iter=38, densify_grad_threshold=1e-6, position_lr_max_steps=90)


def _args_fast():
    a = _args()
    a.max_gaussian_size = 7000
    return a


def _new_log():
    return dict(loss=[], n=[], k=[], densify=0, prune=0, reset=0, grown=0)


def _fresh_model(args, opt):
    torch.manual_seed(0)
    pcd, cams = _scene()
    g = GaussianModel(3, args)
    g.set_inputDim(2 * 6, 6 * 10)
    g.create_from_pcd(pcd, 2.0)
    g.deterministic_surgery = True                     # (single process too: the split samples come from the seeded stream)
    g.bucket_small_numel = 4000                        # 1500 Gaussians: the per-Gaussian tensors get regions of their own, as at 1 M
    g.training_setup(opt)
    return g, cams, [c.original_image for c in cams]


def _rank_schedule(rank, world, port, out_dir, sharded, last):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args, opt = _args_fast(), _opt_fast()
    g, cams, gts = _fresh_model(args, opt)
    log = _new_log()
    _run_pairs(g, cams, gts, opt, args, last, world, rank, 1, log, sharded=sharded)
    sd = g.optimizer.state_dict()                      # (sharded: a collective)
    torch.save({"params": {n: p.detach().cpu() for n, p in g.named_parameters()},
                "state": {k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in sd["state"].items()},
                "stats": {k: getattr(g, k).cpu() for k in ("xyz_gradient_accum", "denom", "max_radii2D")},
                "lag": dict(g.optimizer.lag), "log": log}, os.path.join(out_dir, f"sch{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sharded", [True, False])
def test_two_rank_schedule_stays_rank_identical_through_surgery(tmp_path, sharded):
    """View-parallel over two ranks through >= 3 densify + prune events, an opacity reset, the k-means hook and keypoint growth, on
    the kernels: both ranks end with bit-identical parameters, Adam moments, statistics, N and K (the screen-space gradient every
    rank accumulates is the LAST rank's, train_step.py; seeded split samples, broadcast k-means, checksums after every surgery:
    training.py) -- and the trajectory is the single-process `--batch 2` run's [REF train.py:113-133, 164-192]."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    last = 51
    mp.spawn(_rank_schedule, args=(2, port, str(tmp_path), sharded, last), nprocs=2, join=True)
    r = [torch.load(os.path.join(tmp_path, f"sch{k}.pt"), weights_only=False) for k in range(2)]
    L = r[0]["log"]
    assert L["densify"] >= 3 and L["prune"] >= 3 and L["reset"] >= 1 and L["grown"] >= 1, {k: L[k] for k in ("densify", "prune", "reset", "grown")}
    assert max(L["n"]) > 1500 and L["k"][-1] > 24
    assert L["n"] == r[1]["log"]["n"] and L["k"] == r[1]["log"]["k"] and r[0]["lag"] == r[1]["lag"]
    for k in r[0]["params"]:
        assert torch.equal(r[0]["params"][k], r[1]["params"][k]), k
    for k in r[0]["state"]:
        for kk in ("exp_avg", "exp_avg_sq", "step"):
            assert torch.equal(r[0]["state"][k][kk], r[1]["state"][k][kk]), (k, kk)
    for k in r[0]["stats"]:
        assert torch.equal(r[0]["stats"][k], r[1]["stats"][k]), k
    # ---- the single-process --batch 2 run over the same pairs.  The kernels accumulate with atomics and Adam(eps = 1e-15) amplifies
    # the last bits, so the two trajectories are compared as two runs of ONE implementation are: the same events,
    # N within 2 % and K within 2 after every iteration, the per-step losses (sum of the two ranks' = the batch loss) within 2 %.
    args, opt = _args_fast(), _opt_fast()
    g, cams, gts = _fresh_model(args, opt)
    one = _new_log()
    _run_pairs(g, cams, gts, opt, args, last, 1, 0, 2, one)
    assert {k: one[k] for k in ("densify", "prune", "reset")} == {k: L[k] for k in ("densify", "prune", "reset")}
    n1, n2 = np.array(one["n"], dtype=np.float64), np.array(L["n"], dtype=np.float64)
    assert np.abs(n1 - n2).max() <= 0.02 * n1.max() and np.abs(np.array(one["k"]) - np.array(L["k"])).max() <= 2, (one["n"], L["n"], one["k"], L["k"])
    both = np.array(L["loss"]) + np.array(r[1]["log"]["loss"])
    assert np.abs(both - np.array(one["loss"])).max() <= 0.02 * np.abs(np.array(one["loss"])).max(), np.abs(both - np.array(one["loss"])).max()


# ---- a loop in the REFERENCE's order on the drop-in surfaces ---------------------------------------------------------------------
def _reference_order_iteration(model, cam, gt, it, opt, pipe, bg, extent, cap):
    """One iteration the way the reference sequences it [REF train.py:101-197]: backward, THEN the densification calls on the model,
    THEN `optimizer.step()` / `optimizer.zero_grad(set_to_none=True)` -- only the model's and the optimizer's public methods."""
    from gaussianprediction_amd.loss_ops import l1_ssim_loss
    model.update_learning_rate(it)
    t = torch.from_numpy(cam.time).float().cuda()
    pkg = render(cam, model, pipe, bg, time=t, it=it)
    loss = l1_ssim_loss(pkg["render"], gt, opt.lambda_dssim) + model.get_loss(it)
    loss.backward()
    with torch.no_grad():
        if it < opt.densify_until_iter:
            vis, radii = pkg["visibility_filter"], pkg["radii"]
            model.max_radii2D[vis] = torch.max(model.max_radii2D[vis], radii[vis].float())
            model.add_densification_stats(pkg["viewspace_points"], vis)
            due = it > opt.densify_from_iter and it % opt.densification_interval == 0
            big = 20 if it > opt.opacity_reset_interval else None
            if due and model.get_xyz.shape[0] < cap:
                model.densify(opt.densify_grad_threshold, 0.005, extent, big)
            if it % opt.opacity_reset_interval == 0:
                model.reset_opacity()
            if due:
                model.prune(opt.densify_grad_threshold, 0.005, extent, big)
        model.optimizer.step()
        model.optimizer.zero_grad(set_to_none=True)
    return float(loss)


def test_reference_order_loop_equals_the_harness_order():
    """The advisor's round-4 finding: nothing ran `gaussians.optimizer.step()` AFTER the surgery any more.  In that order the rebuilt
    bucket must keep this iteration's gradient for the tensors that survive (MLP) and pass over the replaced ones, as
    torch.optim.Adam does with .grad None.  Checked against this package's own order (TrainStep: update with the replaced groups
    held, then operate), which is the same computation: same N after every iteration, same step counts, parameters within the
    noise of two runs."""
    args, opt = _args_fast(), _opt_fast()
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    bg = torch.zeros(3, device="cuda")
    last = 34                                           # stage 1: densify at 6, 12, 18 (then the cap), prune every 6, the opacity reset at 30
    out = []
    for order in ("reference", "harness"):
        g, cams, gts = _fresh_model(args, opt)
        ns, losses = [], []
        if order == "reference":
            for it in range(1, last + 1):
                v = (7 * it) % len(cams)
                losses.append(_reference_order_iteration(g, cams[v], gts[v], it, opt, pipe, bg, 2.0, args.max_gaussian_size))
                ns.append(g.get_xyz.shape[0])
        else:
            ts = TrainStep(g, cams, gts, 1, lambda_dssim=opt.lambda_dssim, schedule=True, training_args=opt)
            for it in range(1, last + 1):
                ts.iteration = it
                loss, pkg = ts.step((7 * it) % len(cams), hold=dn.held_groups(g, it, opt))
                with torch.no_grad():
                    dn.track_view(g, pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"])
                    dn.densification_step(g, it, opt, 2.0, max_gaussian_size=args.max_gaussian_size)
                losses.append(float(loss)), ns.append(g.get_xyz.shape[0])
        sd = g.optimizer.state_dict()
        lr_of = {id(p): float(gr["lr"]) for gr in g.optimizer.param_groups for p in gr["params"]}
        out.append(dict(n=ns, loss=losses, lag=dict(g.optimizer.lag), steps={k: float(v["step"]) for k, v in sd["state"].items()},
                        params={n: p.detach().clone() for n, p in g.named_parameters()},
                        lr={n: max(lr_of.get(id(p), 1e-3), 1e-6) for n, p in g.named_parameters()}))
    ref, har = out
    assert ref["n"][-1] > 1500 and len(set(ref["n"])) >= 3
    assert np.abs(np.array(ref["n"]) - np.array(har["n"])).max() <= 0.01 * max(ref["n"]), (ref["n"], har["n"])
    assert ref["lag"] == har["lag"] and ref["lag"].get("xyz", 0) >= 2 and ref["steps"] == har["steps"]
    assert np.abs(np.array(ref["loss"]) - np.array(har["loss"])).max() < 0.02
    if ref["n"] == har["n"]:
        # Same rows: compare element for element, in units of a learning-rate step.  The bars come from a calibration, not from a guess
        # (tools/probe/order_noise.py, profiles/r05_order_noise.txt): two runs of the SAME order end 0.01 - 0.23 lr apart in the median
        # (the MLP weights and the motion feature most: small gradients, Adam's eps = 1e-15), 0.02 - 1.8 lr at the 99th percentile,
        # 0.1 - 23 lr at the maximum -- and the two orders end exactly that far from each other (0.0001 - 0.22 / 0.01 - 1.7 / 0.1 - 9).
        for k, a in ref["params"].items():
            lr, d = ref["lr"][k], (a - har["params"][k]).abs().flatten().float()
            q99 = float(torch.quantile(d[:1_000_000], 0.99)) if d.numel() > 1 else float(d.max())
            assert float(d.median()) <= 0.5 * lr + 1e-7 and q99 <= 4 * lr + 1e-6 and float(d.max()) <= 2 * last * lr + 1e-5, \
                (k, float(d.median()), q99, float(d.max()), lr)
