"""A CPU stand-in for `render()` with the product's result dict, built ONLY from checkers: oracle/deform_oracle.py (the
deformation, pinned by the reference's golden vectors) -> tests/dense_ref.py (dense per-pixel splatting, torch autograd).
Test infrastructure: it lets the `-m "not gpu"` tests drive the HOST side of a whole training iteration --
train_step.TrainStep, the gradient exchange, the fused optimizer's bookkeeping, densify / prune / keypoint growth --
on CPU tensors under gloo, where no HIP kernel can run.  Nothing under gaussianprediction_amd/ imports it; the tests
install it over `train_step.render` / `train_step.l1_ssim_loss` explicitly (`install()`)."""
import math
from types import SimpleNamespace

import torch

from oracle import deform_oracle as do
from dense_ref import dense_render
from host_checkers import torch_l1_ssim


def _keypoint_inputs(pc, nn_):
    """Stage 2/3 inputs of the blend, as plain functions of the parameters (the product computes them with its hash-grid
    weights model and neighbour search, HIP only): exact nearest keypoints in position space, raw weights = a fixed
    smooth function of the position."""
    xyz, kp = pc._xyz.detach().double(), pc.super_gaussians.detach().double()
    idx = torch.cdist(xyz, kp).topk(nn_, dim=-1, largest=False).indices
    g = torch.Generator().manual_seed(11)
    A = torch.randn(3, 2 * nn_, generator=g, dtype=torch.float64)
    return torch.sin(xyz @ A), idx


def host_render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, delta=None, time=None, it=1,
                binning=None):
    a = pc.args
    args = SimpleNamespace(**vars(a))
    args.xyz_freq, args.time_freq = int(pc.xyz_input_dim / 6), pc.time_input_dim // 2
    args.second_stage_iteration = pc.second_stage_iter
    pc.stage_transitions(it)                       # (the hooks forward() would run)
    wait = getattr(pc, "_param_ready_wait", None)  # (sharded optimizer: outstanding all-gathers of updated parameter slices)
    if wait is not None:
        wait()
    P = dict(xyz=pc._xyz, rotation=pc._rotation, scaling=pc._scaling, opacity=pc._opacity, motion_feature=pc.motion_feature)
    raw_w = knn = None
    if it > pc.second_stage_iter:
        P.update(super_gaussians=pc.super_gaussians, super_gaussians_feature=pc.super_gaussians_feature)
        raw_w, knn = _keypoint_inputs(pc, a.nearest_num)
    P = {k: v.double() for k, v in P.items()}
    sd = {k: v.double() for k, v in pc.df_model.named_parameters()}
    t = time.detach().double().reshape(1)
    xyz_t, q, s, o = do.deform_forward(P, sd, t, it, args, raw_w=raw_w, knn_idx=knn)
    n = xyz_t.shape[0]
    screenspace = torch.zeros(n, 3, dtype=torch.float32, requires_grad=True)
    shs = torch.cat([pc._features_dc, pc._features_rest], dim=1).double()
    c = viewpoint_camera
    img, depth, radii, tidx = dense_render(xyz_t, screenspace.double(), shs, None, o, s, q, None, c.world_view_transform.double(),
                                           c.full_proj_transform.double(), c.camera_center.double(), bg_color.double(),
                                           int(c.image_height), int(c.image_width), math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5),
                                           pc.active_sh_degree)
    return {"render": img.float(), "viewspace_points": screenspace, "visibility_filter": radii > 0, "radii": radii,
            "depth": depth.float(), "tidx": tidx}


def host_loss(image, gt, lambda_dssim=0.2, reg_x=None, reg_scale=0.0):
    loss = torch_l1_ssim(image, gt, lambda_dssim)
    return loss if reg_x is None else loss + reg_scale * reg_x.abs().mean()


def install():
    from gaussianprediction_amd import train_step
    train_step.render = host_render
    train_step.l1_ssim_loss = host_loss
