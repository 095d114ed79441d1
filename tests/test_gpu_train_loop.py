"""-m gpu: the reference's training loop and its restart / evaluation paths, call for call, on the drop-in surfaces
(`from gaussian_renderer import render, render_motion, GaussianModel`) -- with the stage thresholds compressed so that 90
iterations cross everything train.py does [REF train.py:36-201, eval.py:226-260]:
warm-up (static) -> stage 1 (per-Gaussian MLP; densify / prune / opacity reset) -> the hook at second_stage_iter + 1 (k-means
keypoints, stage-2 optimizer) -> keypoint growth -> the hook at third_stage_iter + 1 -> stage 3 (hash-grid weights model + kNN
evaluated inside forward) -> checkpoint tuple -> a fresh model sized from it (N_pcd_init / final_kpts_num, create_from_pcd,
training_setup, restore, load_state_dict) that continues to train -> eval.py's restore-without-training_setup and its render loop.
The loss is the reference's own torch code path (its utils/loss_utils.py stays Python on the reference side; the restatement in
oracle/deform_oracle.py is pinned to it by golden vectors): the loop below touches this package only through the reference's API."""
import os
from random import Random
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gaussian_renderer import GaussianModel, render, render_motion  # noqa: E402  (the import-name shims)
from gaussianprediction_amd.cameras import orbit_cameras  # noqa: E402
from gaussianprediction_amd.training import default_training_args  # noqa: E402
from oracle import deform_oracle as do  # noqa: E402

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


def _train(gaussians, cams, opt, args, pipe, background, first_iter, last_iter, rnd, log, batch=1):
    """train.py:76-201 (with --batch > 1 the optimizer steps every `batch` iterations on the sum of their losses)."""
    viewpoint_stack = None
    batch_loss, batch_viewspace_point_tensor, batch_radii, batch_visibility_filter = [], [], [], []
    for iteration in range(first_iter, last_iter + 1):
        gaussians.update_learning_rate(iteration)
        if iteration % 20 == 0:                                   # (every 1000 in the reference)
            gaussians.oneupSHdegree()
        if not viewpoint_stack:
            viewpoint_stack = list(cams)
        viewpoint_cam = viewpoint_stack.pop(rnd.randint(0, len(viewpoint_stack) - 1))
        max_frame = len(cams)
        decay_noise = torch.randn([1], device="cuda") * args.time_noise_ratio / max_frame * (1 - min(1, iteration / args.time_noise_iteration))
        if args.use_time_decay and iteration >= gaussians.second_stage_iter:
            decay_noise = torch.randn([1], device="cuda") * args.time_noise_ratio / max_frame * \
                (1 - min(1, (iteration - gaussians.second_stage_iter) / (args.time_noise_iteration * 2)))
        time_ = torch.from_numpy(viewpoint_cam.time).to(torch.float32).to("cuda") + decay_noise
        render_pkg = render(viewpoint_cam, gaussians, pipe, background, delta=None, time=time_, it=iteration)
        image, viewspace_point_tensor, visibility_filter, radii = render_pkg["render"], render_pkg["viewspace_points"], \
            render_pkg["visibility_filter"], render_pkg["radii"]
        gt_image = viewpoint_cam.original_image.cuda()
        Ll1 = do.l1_loss(image, gt_image)
        loss = (1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - do.ssim(image, gt_image))
        loss += gaussians.get_loss(iteration)
        batch_loss += [loss]
        batch_radii += [radii.unsqueeze(0)]
        batch_visibility_filter += [visibility_filter.unsqueeze(0)]
        batch_viewspace_point_tensor += [viewspace_point_tensor]
        if len(batch_loss) == batch:
            loss_ = torch.stack(batch_loss, dim=0).sum()
            loss_.backward()
            radii = torch.cat(batch_radii, 0).max(dim=0).values
            visibility_filter = torch.cat(batch_visibility_filter).any(dim=0)
            viewspace_point_tensor_grad = torch.zeros_like(viewspace_point_tensor)
            for idx in range(0, len(batch_viewspace_point_tensor)):
                viewspace_point_tensor_grad = viewspace_point_tensor_grad + batch_viewspace_point_tensor[idx].grad
            assert torch.isfinite(viewspace_point_tensor_grad).all()
            batch_loss.clear(), batch_radii.clear(), batch_visibility_filter.clear(), batch_viewspace_point_tensor.clear()
        else:
            log["loss"].append(loss.item()), log["n"].append(gaussians.get_xyz.shape[0]), log["k"].append(gaussians.super_gaussians.shape[0])
            continue
        with torch.no_grad():
            log["loss"].append(loss.item())
            log["n"].append(gaussians.get_xyz.shape[0])
            log["k"].append(gaussians.super_gaussians.shape[0])
            if iteration < opt.densify_until_iter:
                gaussians.max_radii2D[visibility_filter] = torch.max(gaussians.max_radii2D[visibility_filter], radii[visibility_filter])
                gaussians.add_densification_stats(viewspace_point_tensor, visibility_filter)
                if iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0 and gaussians.get_xyz.shape[0] < args.max_gaussian_size:
                    size_threshold = 20 if iteration > opt.opacity_reset_interval else None
                    gaussians.densify(opt.densify_grad_threshold, 0.005, 2.0, size_threshold)
                if iteration % opt.opacity_reset_interval == 0:
                    gaussians.reset_opacity()
                if iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0:
                    size_threshold = 20 if iteration > opt.opacity_reset_interval else None
                    gaussians.prune(opt.densify_grad_threshold, 0.005, 2.0, size_threshold)
            if iteration < args.adaptive_end_iter + gaussians.second_stage_iter and gaussians.super_gaussians.shape[0] < args.max_points + args.adaptive_points_num:
                if gaussians.second_stage:
                    gaussians.max_radii2D[visibility_filter] = torch.max(gaussians.max_radii2D[visibility_filter], radii[visibility_filter])
                    gaussians.add_densification_stats(viewspace_point_tensor, visibility_filter)
                if iteration > args.adaptive_from_iter + gaussians.second_stage_iter and iteration % args.adaptive_interval == 0:
                    if gaussians.new_xyz is not None:
                        gaussians.densification_motion_postfix(gaussians.new_xyz, gaussians.new_motion_feature)
                        gaussians.new_kpts_init()
                    if args.densify_from_grad == "True":
                        gaussians.densify_kpts(opt.densify_grad_threshold, mode="down_sampling")
            if iteration < opt.iterations:
                gaussians.optimizer.step()
                gaussians.optimizer.zero_grad(set_to_none=True)
    return iteration


@pytest.mark.parametrize("batch", [1, 2])
def test_the_reference_training_loop_runs_unchanged(tmp_path, batch):
    torch.manual_seed(0)
    args, opt = _args(), _opt()
    pcd, cams = _scene()
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    background = torch.tensor([0, 0, 0], dtype=torch.float32, device="cuda")
    gaussians = GaussianModel(3, args)
    gaussians.set_inputDim(2 * 6, 6 * 10)
    gaussians.create_from_pcd(pcd, 2.0)                            # (Scene.__init__)
    gaussians.training_setup(opt)
    log = dict(loss=[], n=[], k=[])
    rnd = Random(0)
    last = 75 + (batch - 1)
    _train(gaussians, cams, opt, args, pipe, background, 1, last, rnd, log, batch)
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
    _train(g2, cams, opt, args, pipe, background, first_iter + 1, 90, Random(1), log2, batch)
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
