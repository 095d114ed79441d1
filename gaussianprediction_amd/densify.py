"""The densification calls of the reference's training loop, in the order train.py makes them
[REF train.py:164-192]:

    max_radii2D update + add_densification_stats           (every iteration below densify_until_iter)
    densify (only while N < max_gaussian_size)              -> GaussianModel.densify   (clone + split, statistics reset)
    reset_opacity (every opacity_reset_interval)            -> GaussianModel.reset_opacity
    prune (always, also when densify was skipped)           -> GaussianModel.prune
    keypoint growth in the second stage                     -> GaussianModel.densification_motion_postfix / densify_kpts

The operations themselves are methods of the model (training.py), as in the reference; this module is the loop-side driver.
"""
from __future__ import annotations

import torch


def track_view(model, viewspace_point_tensor, visibility_filter, radii):
    """[REF train.py:166-167]"""
    model.max_radii2D[visibility_filter] = torch.max(model.max_radii2D[visibility_filter], radii[visibility_filter].to(torch.float32))
    model.add_densification_stats(viewspace_point_tensor, visibility_filter)


def held_groups(model, iteration, opt, white_background=False):
    """Names of the optimizer groups whose Adam update the REFERENCE skips on this iteration: its loop runs densify / prune /
    reset_opacity between backward and optimizer.step() [REF train.py:164-197]; they replace the per-Gaussian Parameters (prune
    always does, whatever its mask [REF scene/gaussian_model.py:547-585]; reset_opacity the opacity tensor [REF :526-545]), the new
    tensors' .grad is None and torch.optim.Adam passes over them: no update from this iteration's gradient, step count one behind.
    This package's harness updates first and operates afterwards, so it asks for the same outcome up front:
        ts.step(view, hold=densify.held_groups(model, it, opt))
    Not predicted (it depends on this iteration's gradients): the keypoint tensors on an iteration whose keypoint growth actually
    adds keypoints [REF train.py:179-192]."""
    if not iteration < opt.densify_until_iter:
        return ()
    names = set()
    if iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0:
        names.update(model._per_gaussian().keys())
    if iteration % opt.opacity_reset_interval == 0 or (white_background and iteration == opt.densify_from_iter):
        names.add("opacity")
    have = {g["name"] for g in model.optimizer.param_groups} if model.optimizer is not None else set()
    return tuple(sorted(names & have))


def densification_step(model, iteration, opt, scene_extent, max_gaussian_size=200_000, white_background=False, generator=None):
    """[REF train.py:169-177]: returns (n_cloned, n_split_sources, n_pruned), None where the call was not due."""
    n_clone = n_src = n_pruned = None
    due = iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0
    size_threshold = 20 if iteration > opt.opacity_reset_interval else None
    if due and model.get_xyz.shape[0] < max_gaussian_size:
        n_clone, n_src = model.densify(opt.densify_grad_threshold, 0.005, scene_extent, size_threshold, generator=generator)
    if iteration % opt.opacity_reset_interval == 0 or (white_background and iteration == opt.densify_from_iter):
        model.reset_opacity()
    if due:
        n_pruned = model.prune(opt.densify_grad_threshold, 0.005, scene_extent, size_threshold)
    return n_clone, n_src, n_pruned


def keypoint_growth_step(model, iteration, opt, args, visibility_filter=None, radii=None, viewspace_point_tensor=None):
    """[REF train.py:179-192]: second-stage statistics, adoption of the teacher's candidates, gradient-driven down-sampling."""
    off = model.second_stage_iter
    if not (iteration < args.adaptive_end_iter + off and model.super_gaussians.shape[0] < args.max_points + args.adaptive_points_num):
        return False
    if model.second_stage and viewspace_point_tensor is not None:
        track_view(model, viewspace_point_tensor, visibility_filter, radii)
    grown = False
    if iteration > args.adaptive_from_iter + off and iteration % args.adaptive_interval == 0:
        k0 = model.super_gaussians.shape[0]
        if model.new_xyz is not None:          # the teacher (densify_from_teaching) proposed keypoints
            model.densification_motion_postfix(model.new_xyz, model.new_motion_feature)
            model.new_kpts_init()
        if str(getattr(args, "densify_from_grad", "True")) == "True":
            model.densify_kpts(opt.densify_grad_threshold, mode="down_sampling")
        grown = model.super_gaussians.shape[0] != k0
    return grown
