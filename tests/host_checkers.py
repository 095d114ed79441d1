"""Checkers the tests run BESIDE the product -- plain-torch restatements of steps whose only product implementation is a
HIP kernel.  Test infrastructure: nothing under gaussianprediction_amd/ imports this module; two of the functions are
installed through explicit seams of the package (`FusedAdam.host_step`, `training.host_fps`) so that the -m "not gpu" tests can
drive the host-side bookkeeping (sharding, optimizer-state surgery, keypoint growth) on CPU tensors."""
import torch

from oracle import deform_oracle as do


@torch.no_grad()
def adam_host_step(opt, step_no, zero_grad, keep_ids):
    """torch.optim.Adam's update rule (amsgrad off, no weight decay) over a FusedAdam's launch table
    [REF scene/gaussian_model.py:472, train.py:196-197]."""
    b1, b2 = opt.betas
    bc1, bc2 = 1 - b1 ** step_no, 1 - b2 ** step_no
    for (g, p, off, m, v) in opt.items:
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
