"""Train-step harness: what the reference's loop does around the hot
path for one iteration [REF train.py:101-133, 196-197]:
    render -> 0.8*L1 + 0.2*(1-SSIM_11x11) + 1e-5*mean|motion feature| -> backward -> Adam(eps=1e-15).
Render, deformation, loss (fused L1+SSIM) and optimizer (fused Adam + gradient zeroing) all run on
this package's HIP kernels; `fused=False` switches loss and optimizer to the plain-torch restatement of
the reference (used by the tests as the checker).
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from .dist import FlatGradBucket, OverlappedGradReducer
from . import grad_sink
from .loss_ops import FusedAdam, add_l1_mean, l1_ssim_loss
from .renderer import render


def _gauss_window(channels, device, size=11, sigma=1.5):
    import math
    g = torch.tensor([math.exp(-(x - size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(size)], device=device)
    g = (g / g.sum()).unsqueeze(1)
    return (g @ g.t()).unsqueeze(0).unsqueeze(0).expand(channels, 1, size, size).contiguous()


def l1_loss(a, b):                       # [REF utils/loss_utils.py:54-55]
    return torch.abs(a - b).mean()


def ssim(img1, img2, window):            # [REF utils/loss_utils.py:70-100]
    C = img1.shape[-3]
    i1, i2 = img1.unsqueeze(0), img2.unsqueeze(0)
    pad = window.shape[-1] // 2
    mu1 = F.conv2d(i1, window, padding=pad, groups=C)
    mu2 = F.conv2d(i2, window, padding=pad, groups=C)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = F.conv2d(i1 * i1, window, padding=pad, groups=C) - mu1_sq
    s2 = F.conv2d(i2 * i2, window, padding=pad, groups=C) - mu2_sq
    s12 = F.conv2d(i1 * i2, window, padding=pad, groups=C) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))).mean()


class TrainStep:
    """One optimisation step over one view per rank (view-parallel when world_size > 1)."""

    # capacity mode of the rasterizer's binning stage (include/gp_hip.h, gp_raster_settings.binning_capacity): the host
    # never waits for R.  Protocol: the first `len(cameras)` steps run in exact mode and report R through the status word;
    # afterwards the capacity is SPEC_MARGIN x the largest R seen, every step writes {R, overflow} into one of SPEC_SLOTS
    # status slots, Adam takes the overflow word as its skip flag, and the slot is read back (pinned, asynchronous) when it
    # comes round again SPEC_SLOTS steps later: the high-water mark follows the scene, and a frame that overflowed (its
    # update was skipped on the device) is repeated in exact mode.
    SPEC_SLOTS = 4
    SPEC_MARGIN = 1.1
    SPEC_PAD = 4096

    def __init__(self, pc, cameras, gt_images, iteration, lambda_dssim=0.2, lrs=None, group=None, fused=True, speculative=False,
                 overlap_sh_adam=False):
        self.pc, self.cameras, self.gt, self.iteration = pc, cameras, gt_images, iteration
        self.lambda_dssim = lambda_dssim
        self.group = group
        self.fused = fused
        dev = pc.get_xyz.device
        self.speculative = bool(speculative) and fused and dev.type == "cuda"
        if self.speculative:
            K = self.SPEC_SLOTS
            self._status = torch.zeros(K, 2, dtype=torch.int32, device=dev)
            self._status_host = torch.zeros(K, 2, dtype=torch.int32).pin_memory()
            self._events = [None] * K
            self._slot_view = [None] * K            # view rendered by the step that last used the slot
            self._slot_spec = [False] * K           # ... and whether it ran in capacity mode
            self._r_max, self._n_steps, self.redone = 0, 0, 0
            if os.environ.get("GP_SPEC_MARGIN"):     # test hook: a margin < 1 forces overflows (and the redo protocol)
                self.SPEC_MARGIN, self.SPEC_PAD = float(os.environ["GP_SPEC_MARGIN"]), 0
        # Optional (off: measured 1.83 -> 1.90 ms on the bench): Adam for the SH coefficients (3/4 of all parameter bytes) on
        # a second stream.  Their gradients are final as soon as the rasterizer backward has run and their values are not
        # read again before the next rasterizer forward, so the update can overlap the deformation backward of this step and
        # the deformation forward of the next -- but those kernels share the HBM with it and queue behind its 2800
        # workgroups, which costs more than the overlap hides.  Single-process only.
        self.overlap_sh_adam = bool(overlap_sh_adam) and fused and dev.type == "cuda"
        self._side = torch.cuda.Stream(device=dev) if self.overlap_sh_adam else None
        self._sh_early = False
        self._sink_cb = None
        self.bg = torch.zeros(3, device=dev)          # black background [REF train.py:59]
        self.pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
        self.window = _gauss_window(3, dev)
        # camera times live on the device: a per-step H2D copy from pageable memory would be a host sync
        self.times = [torch.from_numpy(c.time).to(torch.float32).to(dev) for c in cameras]
        lr = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=2.5e-3 / 20, opacity=0.05, scaling=5e-3, rotation=1e-3, mfeature=8e-4,
                  kpts=8e-4, mlp=8e-4, hash=5e-3)   # [REF arguments/__init__.py:74-90]
        if lrs:
            lr.update(lrs)
        self.lr = lr
        self._build_optimizer()

    def _groups(self):
        """Parameter groups per training stage, as the reference builds them
        [REF scene/gaussian_model.py:394-411 (stage 3), 413-430 (stage 2), 432-451 (stage 1)]."""
        pc, lr, iteration = self.pc, self.lr, self.iteration
        g_gauss = [
            {"params": [pc._xyz], "lr": lr["xyz"], "name": "xyz"},
            {"params": [pc._features_dc], "lr": lr["f_dc"], "name": "f_dc"},
            {"params": [pc._features_rest], "lr": lr["f_rest"], "name": "f_rest"},
            {"params": [pc._opacity], "lr": lr["opacity"], "name": "opacity"},
            {"params": [pc._scaling], "lr": lr["scaling"], "name": "scaling"},
            {"params": [pc._rotation], "lr": lr["rotation"], "name": "rotation"},
        ]
        g_mlp = [{"params": list(pc.df_model.parameters()), "lr": lr["mlp"], "name": "df_mlp"}]
        g_kp = []
        if hasattr(pc, "super_gaussians"):
            g_kp = [{"params": [pc.super_gaussians], "lr": lr["kpts"], "name": "s_xyz"},
                    {"params": [pc.super_gaussians_feature], "lr": lr["kpts"], "name": "s_motion_feature"}]
            if getattr(pc, "weights_model", None) is not None and pc.raw_weights is None:   # [REF :402,421 "weight_mlp"]
                g_kp.append({"params": list(pc.weights_model.parameters()), "lr": lr["hash"], "name": "weight_mlp"})
        if iteration <= pc.second_stage_iter:                      # stage 1
            return g_gauss + g_mlp + [{"params": [pc.motion_feature], "lr": lr["mfeature"], "name": "motion_feature"}]
        if iteration <= pc.third_stage_iter:                       # stage 2: keypoints + MLP only
            return g_kp + g_mlp
        return g_gauss + g_kp + g_mlp                              # stage 3: everything except the per-Gaussian feature

    def _build_optimizer(self):
        pc = self.pc
        groups = self._groups()
        optimized = {id(p) for g in groups for p in g["params"]}
        for p in pc.parameters():                                  # the reference computes (and ignores) these grads
            p.requires_grad_(id(p) in optimized)
        self.bucket = FlatGradBucket([p for g in groups for p in g["params"]])
        self.reducer = OverlappedGradReducer(self.bucket, self.group)
        if getattr(self, "overlap_sh_adam", False) and self._sink_cb is None:
            self._armed = False
            self._ev_bwd, self._ev_sh = torch.cuda.Event(), torch.cuda.Event()
            self._sink_cb = grad_sink.register_callback(self._on_sink)
        if self.fused:
            self.optimizer = FusedAdam(groups, self.bucket, eps=1e-15)
        else:
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15, foreach=True)   # [REF scene/gaussian_model.py:472]

    # ---- optimizer-state surgery (densify / prune, gaussianprediction_amd/densify.py) ---------------------------
    def adam_moments(self):
        """{id(param): (exp_avg, exp_avg_sq)} of the current optimizer."""
        self.wait_side()
        if self.fused:
            return {id(p): (m, v) for _, p, _, m, v in self.optimizer.items}
        return {id(p): (st["exp_avg"], st["exp_avg_sq"]) for p, st in self.optimizer.state.items() if "exp_avg" in st}

    def rebuild_optimizer(self, carried):
        """After the model's per-Gaussian Parameters were replaced: new bucket + optimizer; `carried` maps id(new param)
        to its (exp_avg, exp_avg_sq); everything else keeps its moments; the step count is preserved."""
        old = self.adam_moments()
        old_steps = self.optimizer.step_count if self.fused else None
        old_torch_state = None if self.fused else {id(p): st for p, st in self.optimizer.state.items()}
        self.reducer.close()
        grad_sink.forget_all()
        self._build_optimizer()
        if self.fused:
            self.optimizer.step_count = old_steps
            for _, p, _, m, v in self.optimizer.items:
                src = carried.get(id(p)) or old.get(id(p))
                if src is not None and src[0].shape == m.shape:
                    m.copy_(src[0]); v.copy_(src[1])
        else:
            for g in self.optimizer.param_groups:
                for p in g["params"]:
                    src = carried.get(id(p))
                    if src is not None:
                        self.optimizer.state[p] = {"step": torch.tensor(0.0), "exp_avg": src[0].clone(), "exp_avg_sq": src[1].clone()}
                    elif id(p) in old_torch_state:
                        self.optimizer.state[p] = old_torch_state[id(p)]

    def wait_side(self):
        """Make the current stream wait for the side-stream SH update (before anything but render() touches the SH
        parameters, their gradients or their Adam moments)."""
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def _on_sink(self, p):
        """grad_sink callback (runs on the autograd thread, inside loss.backward()): the rasterizer backward has written the
        SH gradients -> launch their Adam update on the side stream, behind everything enqueued so far."""
        if not self._armed or p is not self.pc._features_rest:
            return
        self._armed = False
        self._ev_bwd.record()                                   # the autograd thread's current stream = the forward's stream
        self._side.wait_event(self._ev_bwd)
        self.optimizer.step(zero_grad=True, keep_grad=self._keep, skip_flag=self._skip_flag,
                            only=(self.pc._features_dc, self.pc._features_rest), stream=self._side, advance=False)
        self._ev_sh.record(self._side)
        self.pc._param_ready_event = self._ev_sh                # render() waits for it right before the rasterizer call
        self._sh_early = True

    def loss_of(self, image, gt):
        if self.fused:
            loss = l1_ssim_loss(image, gt, self.lambda_dssim)
            if self.iteration >= self.pc.args.jointly_iteration:            # [REF scene/gaussian_model.py:174-178]
                feat = self.pc.super_gaussians_feature if self.iteration > self.pc.second_stage_iter else self.pc.motion_feature
                return add_l1_mean(loss, feat, 1.0e-5)
            return loss
        else:
            Ll1 = l1_loss(image, gt)
            loss = (1.0 - self.lambda_dssim) * Ll1 + self.lambda_dssim * (1.0 - ssim(image, gt, self.window))
        return loss + self.pc.get_loss(self.iteration)

    def step(self, view_index: int):
        if not self.speculative:
            return self._step(view_index, None, None)
        K = self.SPEC_SLOTS
        slot = self._n_steps % K
        if self._events[slot] is not None:           # the step that used this slot K steps ago: long finished
            self._events[slot].synchronize()
            r, overflow = int(self._status_host[slot, 0]), int(self._status_host[slot, 1])
            self._r_max = max(self._r_max, r)
            if overflow and self._slot_spec[slot]:   # its Adam update was skipped on the device: repeat the frame, exactly
                self.redone += 1
                if self.fused:
                    self.optimizer.step_count -= 1
                self._events[slot] = None
                self._run_slot(slot, self._slot_view[slot], exact=True)
                self._n_steps += 1
                return self.step(view_index)
        exact = self._n_steps < len(self.cameras) + K or self._r_max == 0   # until every view's R has been read back
        out = self._run_slot(slot, view_index, exact)
        self._n_steps += 1
        return out

    def _run_slot(self, slot, view_index, exact):
        status = self._status[slot]
        capacity = 0 if exact else (int(self._r_max * self.SPEC_MARGIN) + self.SPEC_PAD)
        out = self._step(view_index, (capacity, status), None if exact else status[1:2])
        self._status_host[slot].copy_(status, non_blocking=True)
        ev = self._events[slot] or torch.cuda.Event()
        ev.record()
        self._events[slot], self._slot_view[slot], self._slot_spec[slot] = ev, view_index, not exact
        return out

    def _step(self, view_index: int, binning, skip_flag):
        cam = self.cameras[view_index % len(self.cameras)]
        gt = self.gt[view_index % len(self.gt)]
        time = self.times[view_index % len(self.cameras)]
        keep = ()
        if self.fused and not self.pipe.convert_SHs_python and self.iteration > self.pc.third_stage_iter:
            # the per-Gaussian gradients each have exactly one producer kernel that writes the whole tensor (SH: rasterizer
            # backward; xyz / rotation: blend backward; scaling / opacity: activation backward): their zeroing pass is
            # skipped and the next backward overwrites instead of accumulating
            keep = (self.pc._features_dc, self.pc._features_rest, self.pc._xyz, self.pc._rotation, self.pc._scaling,
                    self.pc._opacity)
        if self.overlap_sh_adam and keep and not self.reducer.enabled:
            self._armed, self._keep, self._skip_flag = True, keep, skip_flag
        pkg = render(cam, self.pc, self.pipe, self.bg, time=time, it=self.iteration, binning=binning)
        loss = self.loss_of(pkg["render"], gt)
        loss.backward()                              # hooks start the all-reduce of each large gradient as it completes
        self._armed = False
        self.reducer.finish()                        # SUM over views == the reference's --batch semantics
        if skip_flag is not None and self.reducer.enabled:
            # one rank's overflow invalidates the summed gradient: every rank must skip (and later repeat) this step
            torch.distributed.all_reduce(skip_flag, op=torch.distributed.ReduceOp.MAX, group=self.group)
        if self.fused:
            if self._sh_early:       # the SH tensors were updated on the side stream during the backward
                self._sh_early = False
                self.optimizer.step(zero_grad=True, keep_grad=keep, skip_flag=skip_flag,
                                    exclude=(self.pc._features_dc, self.pc._features_rest))
            else:
                self.optimizer.step(zero_grad=True, keep_grad=keep, skip_flag=skip_flag)
        else:
            self.optimizer.step()
            self.bucket.zero()
        return loss.detach(), pkg
