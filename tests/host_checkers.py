"""Checkers the tests run BESIDE the product -- plain-torch restatements of steps whose only product implementation is a
HIP kernel.  Test infrastructure: nothing under gaussianprediction_amd/ imports this module; two of the functions are
installed through explicit seams of the package (`FusedAdam.host_step`, `training.host_fps`) so that the -m "not gpu" tests can
drive the host-side bookkeeping (sharding, optimizer-state surgery, keypoint growth) on CPU tensors."""
import torch

from oracle import deform_oracle as do


@torch.no_grad()
def adam_host_step(opt, step_no, zero_grad, keep_ids, active=None):
    """torch.optim.Adam's update rule (amsgrad off, no weight decay) over a FusedAdam's launch table, every tensor with its own
    step count [REF scene/gaussian_model.py:472, train.py:196-197]; `active`: which entries take part (None = all)."""
    b1, b2 = opt.betas
    for k, ((g, p, off, m, v), st) in enumerate(zip(opt.items, opt.item_steps(step_no))):
        if active is not None and not active[k]:
            continue
        bc1, bc2 = 1 - b1 ** st, 1 - b2 ** st
        grad = opt.bucket.flat[off:off + p.numel()].view_as(p)
        m.mul_(b1).add_(grad, alpha=1 - b1)
        v.mul_(b2).addcmul_(grad, grad, value=1 - b2)
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(opt.eps)
        p.addcdiv_(m, denom, value=-float(g["lr"]) / bc1)
    if zero_grad:
        opt.bucket.flat.zero_()


def fps_host(x, m):
    """Iterative furthest-point sampling from point 0, first maximum on ties [REF utils/fps.py:71-88]."""
    n = x.shape[0]
    x = x.detach().cpu().to(torch.float32)
    idx = torch.zeros(m, dtype=torch.int64)
    dist = torch.full((n,), 1e10)
    for j in range(1, m):
        d = ((x - x[idx[j - 1]]) ** 2).sum(-1)
        dist = torch.minimum(dist, d)
        idx[j] = int(torch.argmax(dist))
    return idx


def install():
    """Install the two seams (idempotent)."""
    from gaussianprediction_amd import training
    from gaussianprediction_amd.loss_ops import FusedAdam
    FusedAdam.host_step = staticmethod(adam_host_step)
    training.host_fps = fps_host
    from gaussianprediction_amd import dist as gdist
    gdist.host_sh_factor_gradient = sh_factor_gradient_host


def torch_l1_ssim(image, gt, lambda_dssim=0.2):
    """(1 - l) L1 + l (1 - SSIM_11x11) [REF train.py:105-108, utils/loss_utils.py:54-100] in torch ops (autograd)."""
    return (1.0 - lambda_dssim) * do.l1_loss(image, gt) + lambda_dssim * (1.0 - do.ssim(image, gt))


def weights_model_unfused(model, xyz, perm=None):
    """The weights model as the stand-alone encoding kernel + three library GEMMs: the cross-check of the fused kernel."""
    from gaussianprediction_amd.weights_ops import MLP_FLOATS, _HashGridEncode
    p = model.params
    w1, w2, w3 = p[0:4096].view(64, 64), p[4096:8192].view(64, 64), p[8192:MLP_FLOATS].view(16, 64)
    table = p[MLP_FLOATS:].view(-1, 4)
    feat = _HashGridEncode.apply(xyz, table, model.cfg, perm)
    h = torch.relu(feat @ w1.t())
    h = torch.relu(h @ w2.t())
    return (h @ w3.t())[:, :model.n_output_dims]


class TorchTrainStep:
    """The reference's iteration in plain torch around this package's render(): render -> torch L1+SSIM + regulariser ->
    backward -> torch.optim.Adam(eps=1e-15) over the model's own groups [REF train.py:101-133, 196-197]."""

    def __init__(self, pc, cameras, gts, iteration, training_args=None, lambda_dssim=0.2):
        from types import SimpleNamespace
        from gaussianprediction_amd.training import default_training_args
        self.pc, self.cameras, self.gts, self.iteration, self.lam = pc, cameras, gts, iteration, lambda_dssim
        pc.setup_for_iteration(training_args or default_training_args(), iteration)     # bucket (.grad views) + group table
        self.opt = torch.optim.Adam([{"params": g["params"], "lr": g["lr"], "name": g["name"]} for g in pc.optimizer.param_groups],
                                    lr=0.0, eps=1e-15)
        dev = pc.get_xyz.device
        self.bg = torch.zeros(3, device=dev)
        self.pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
        self.times = [torch.from_numpy(c.time).to(torch.float32).to(dev) for c in cameras]

    def step(self, v):
        from gaussianprediction_amd.renderer import render
        pkg = render(self.cameras[v], self.pc, self.pipe, self.bg, time=self.times[v], it=self.iteration)
        loss = torch_l1_ssim(pkg["render"], self.gts[v], self.lam) + self.pc.get_loss(self.iteration)
        loss.backward()
        self.opt.step()
        self.pc.bucket.zero()
        return loss.detach(), pkg


def psnr(img, gt):
    """[REF utils/image_utils.py:18-20] on a [3,H,W] pair: per-channel 20 log10(1 / sqrt(mse)), averaged as train.py:267 does."""
    mse = ((img - gt) ** 2).reshape(img.shape[0], -1).mean(1)
    return float((20.0 * torch.log10(1.0 / torch.sqrt(mse))).mean())


class DenseRefTrainer:
    """The reference's training iteration [REF train.py:101-133, 196-197] with NOTHING of this package in it: torch restatement
    of GaussianModel.forward (oracle/deform_oracle.py, pinned by the reference's golden vectors) -> tests/dense_ref.py (every pixel
    looks at every Gaussian, plain argsort, float64) -> torch L1 + SSIM + regulariser -> torch.autograd -> torch.optim.Adam(eps=1e-15)
    over the reference's parameter groups.  The independent end of the convergence comparison."""

    STAGE_GROUPS = {1: ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "df_mlp", "motion_feature"),
                    3: ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "s_xyz", "s_motion_feature", "df_mlp")}
    KEYS = {"xyz": "xyz", "f_dc": "features_dc", "f_rest": "features_rest", "opacity": "opacity", "scaling": "scaling", "rotation": "rotation",
            "motion_feature": "motion_feature", "s_xyz": "super_gaussians", "s_motion_feature": "super_gaussians_feature"}

    def __init__(self, P, sd, args, cameras, gts, raw_w, knn_idx, sh_degree=3, lambda_dssim=0.2):
        import math
        self.P = {k: v.detach().double().clone().requires_grad_(True) for k, v in P.items()}
        self.sd = {k: v.detach().double().clone().requires_grad_(True) for k, v in sd.items()}
        self.args, self.cams, self.lam, self.sh_degree = args, cameras, lambda_dssim, sh_degree
        self.gts = [g.detach().double().cpu() for g in gts]
        self.raw_w, self.knn_idx = raw_w.detach().double().cpu(), knn_idx.detach().cpu()
        self.views = [dict(V=c.world_view_transform.detach().cpu().double(), Pm=c.full_proj_transform.detach().cpu().double(),
                           campos=c.camera_center.detach().cpu().double(), tfx=math.tan(c.FoVx * 0.5), tfy=math.tan(c.FoVy * 0.5),
                           H=int(c.image_height), W=int(c.image_width), t=torch.tensor(c.time, dtype=torch.float64).reshape(1))
                      for c in cameras]
        self.opt, self.iteration = None, None

    def set_stage(self, iteration, lrs):
        """New optimizer, as the reference's training*_setup calls create one [REF scene/gaussian_model.py:394-472]; lrs by group name."""
        stage = 1 if iteration <= self.args.second_stage_iteration else 3
        groups = []
        for name in self.STAGE_GROUPS[stage]:
            params = list(self.sd.values()) if name == "df_mlp" else [self.P[self.KEYS[name]]]
            groups.append({"params": params, "lr": float(lrs[name]), "name": name})
        self.opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        self.iteration = iteration

    def render(self, v, iteration=None):
        from dense_ref import dense_render
        it = self.iteration if iteration is None else iteration
        c = self.views[v]
        P = self.P
        xyz, q, s, o = do.deform_forward(P, self.sd, c["t"], it, self.args, raw_w=self.raw_w, knn_idx=self.knn_idx)
        shs = torch.cat([P["features_dc"], P["features_rest"]], dim=1)
        img, _, _, _ = dense_render(xyz, torch.zeros(xyz.shape[0], 3, dtype=torch.float64), shs, None, o, s, q, None, c["V"], c["Pm"],
                                    c["campos"], torch.zeros(3, dtype=torch.float64), c["H"], c["W"], c["tfx"], c["tfy"], self.sh_degree)
        return img

    def step(self, v):
        img = self.render(v)
        feat = self.P["super_gaussians_feature"] if self.iteration > self.args.second_stage_iteration else self.P["motion_feature"]
        loss = torch_l1_ssim(img, self.gts[v], self.lam)
        if self.iteration >= self.args.jointly_iteration:                 # [REF scene/gaussian_model.py:174-178]
            loss = loss + 1.0e-5 * feat.abs().mean()
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        self.opt.step()
        self.last_image = img.detach()
        return float(loss.detach())

    def named_parameters(self):
        """(name of the matching GaussianModel attribute / df_model key, tensor) of everything the current optimizer updates."""
        attr = {"xyz": "_xyz", "features_dc": "_features_dc", "features_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
                "rotation": "_rotation", "motion_feature": "motion_feature", "super_gaussians": "super_gaussians",
                "super_gaussians_feature": "super_gaussians_feature"}
        out = []
        for g in self.opt.param_groups:
            if g["name"] == "df_mlp":
                out += [("df_model." + k, self.sd[k]) for k in self.sd]
            else:
                out.append((attr[self.KEYS[g["name"]]], self.P[self.KEYS[g["name"]]]))
        return out

    def adam_state(self, p):
        st = self.opt.state.get(p, {})
        if "exp_avg" not in st:
            return torch.zeros_like(p), torch.zeros_like(p), 0
        return st["exp_avg"], st["exp_avg_sq"], int(float(st["step"]))


def sh_factor_gradient_host(factors, degree, g_dc, g_rest):
    """gp_sh_factor_gradient in torch ops (CPU tensors of the gloo tests): the sum over the views, in order, of Y_k(dir) x dL/dRGB."""
    from gaussianprediction_amd.dist import sh_basis
    world, n, _ = factors.shape
    nc = (degree + 1) ** 2
    acc = torch.zeros(n, 16, 3, dtype=factors.dtype)
    for v in range(world):
        acc[:, :nc] += sh_basis(factors[v, :, 3:6], degree).unsqueeze(2) * factors[v, :, 0:3].unsqueeze(1)
    g_dc.copy_(acc[:, 0:1].reshape(g_dc.shape))
    g_rest.copy_(acc[:, 1:].reshape(g_rest.shape))
