"""render() / render_motion(): the reference's L1 boundary [REF gaussian_renderer/__init__.py:18-191],
same signature and result dict, on this package's rasterizer and GaussianModel."""
from __future__ import annotations

import math

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

_C0, _C1 = 0.28209479177387814, 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435)


def _sh_to_rgb_python(deg, feats, dirs):
    """`pipe.convert_SHs_python` fallback: SH -> RGB in torch (feats [N,16,3], unit dirs [N,3]);
    same basis/constants as [REF utils/sh_utils.py:57-112], + 0.5 and clamp as [REF gaussian_renderer/__init__.py:86-91]."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = _C0 * feats[:, 0]
    if deg > 0:
        res = res - _C1 * y * feats[:, 1] + _C1 * z * feats[:, 2] - _C1 * x * feats[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = res + _C2[0] * xy * feats[:, 4] + _C2[1] * yz * feats[:, 5] + _C2[2] * (2 * zz - xx - yy) * feats[:, 6] + \
            _C2[3] * xz * feats[:, 7] + _C2[4] * (xx - yy) * feats[:, 8]
    if deg > 2:
        res = res + _C3[0] * y * (3 * xx - yy) * feats[:, 9] + _C3[1] * xy * z * feats[:, 10] + \
            _C3[2] * y * (4 * zz - xx - yy) * feats[:, 11] + _C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * feats[:, 12] + \
            _C3[4] * x * (4 * zz - xx - yy) * feats[:, 13] + _C3[5] * z * (xx - yy) * feats[:, 14] + \
            _C3[6] * x * (xx - 3 * yy) * feats[:, 15]
    return torch.clamp_min(res + 0.5, 0.0)


def _settings(viewpoint_camera, pc, bg_color, scaling_modifier, binning=None, sh_ready_event=None, visible_out=None, raw_activations=False):
    return GaussianRasterizationSettings(
        visible_out=visible_out,
        raw_activations=bool(raw_activations),
        binning_capacity=int(binning[0]) if binning else 0,
        binning_status=binning[1] if binning else None,
        depth_key_bits=int(binning[2][0]) if (binning and len(binning) > 2 and binning[2]) else 0,
        depth_key_base=int(binning[2][1]) if (binning and len(binning) > 2 and binning[2]) else 0,
        depth_key_range=binning[3] if (binning and len(binning) > 3) else None,
        sh_ready_event=sh_ready_event,
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5),
        tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
    )


def _wait_params(pc):
    """A training harness may update parameters on a second stream (TrainStep: the SH coefficients); the event it leaves on
    the model orders every render behind that update."""
    ev = getattr(pc, "_param_ready_event", None)
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)
    wait = getattr(pc, "_param_ready_wait", None)     # (sharded optimizer: the all-gather of the SH coefficients)
    if wait is not None:
        wait()


def _screenspace_points(pc):
    """Zero tensor whose .grad receives the 2D (screen-space) mean gradients [REF :27-31].  The reference builds
    `zeros_like(xyz, requires_grad=True) + 0` and retain_grad()s it every frame: two [N,3] kernels, and autograd CLONES the
    incoming gradient into a retained non-leaf's .grad (a third pass over [N,3]).  Here: a fresh LEAF over one cached block of
    zeros (nothing ever writes its values; the rasterizer only routes a gradient through it) -- `.grad` is populated the same
    way and takes the backward's tensor without a copy."""
    xyz = pc.get_xyz
    z = getattr(pc, "_screenspace_zeros", None)
    if z is None or z.shape != xyz.shape or z.device != xyz.device or z.dtype != xyz.dtype:
        z = torch.zeros_like(xyz, requires_grad=False).detach()
        pc._screenspace_zeros = z
    return z.detach().requires_grad_(True)


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None, delta=None,
           time=None, it=1, binning=None):
    """Render the scene.  Background tensor (bg_color) must be on the GPU.
    `binning` (extension, not in the reference signature): (capacity, status[, (depth_key_bits, depth_key_base) | None[, depth_key_range]])
    of the rasterizer's capacity mode / depth-key speculation -- see GaussianRasterizationSettings.binning_capacity, .depth_key_bits."""
    screenspace_points = _screenspace_points(pc)
    cov3D_precomp = None
    raw_act = False
    if time is None:
        means3D = pc.get_xyz + delta if delta is not None else pc.get_xyz
        opacity = pc.get_opacity
        if getattr(pipe, "compute_cov3D_python", False):
            cov3D_precomp, scales, rotations = pc.get_covariance(scaling_modifier), None, None
        else:
            scales, rotations = pc.get_scaling, pc.get_rotation
    else:
        # (a pass without autograd over a model that offers it: the raw scales / opacities go to the rasterizer as they are)
        if not torch.is_grad_enabled() and getattr(pc, "raw_activations_ok", False):
            means3D, rotations, scales, opacity = pc(time, it, raw_activations=True)[:4]
            raw_act = bool(getattr(pc, "_forward_raw", False))
        else:
            means3D, rotations, scales, opacity = pc(time, it)
    shs, shs_rest, colors_precomp = None, None, None
    # (after the deformation was enqueued: it does not read the tensors a harness updates late.)  The SH coefficients are read by
    # the rasterizer alone, and only in front of its composite: a view-parallel harness hands over an EVENT for their all-gather
    # (`_param_late_event`) instead of making this stream wait for it here -- when the two SH parameters go to the kernels as they are.
    split_sh = (override_color is None and not getattr(pipe, "convert_SHs_python", False) and hasattr(pc, "_features_dc")
                and hasattr(pc, "_features_rest") and pc._features_rest.shape[1] == 15)
    late = getattr(pc, "_param_late_event", None) if split_sh else None
    sh_ready = late() if late is not None else None
    if late is None:
        _wait_params(pc)
    vis = torch.empty(means3D.shape[0], dtype=torch.uint8, device=means3D.device)     # filled by the projection kernel (radii > 0)
    rasterizer = GaussianRasterizer(raster_settings=_settings(viewpoint_camera, pc, bg_color, scaling_modifier, binning, sh_ready, vis, raw_act))
    if override_color is None:
        if getattr(pipe, "convert_SHs_python", False):
            base = means3D.detach() if time is not None else pc.get_xyz + (delta if delta is not None else 0)
            d = base - viewpoint_camera.camera_center[None]
            colors_precomp = _sh_to_rgb_python(pc.active_sh_degree, pc.get_features, d / d.norm(dim=1, keepdim=True))
        elif hasattr(pc, "_features_dc") and hasattr(pc, "_features_rest") and pc._features_rest.shape[1] == 15:
            shs, shs_rest = pc._features_dc, pc._features_rest       # no per-frame cat (get_features)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color
    rendered_image, radii, depth, tidx = rasterizer(means3D=means3D, means2D=screenspace_points, shs=shs,
                                                    colors_precomp=colors_precomp, opacities=opacity, scales=scales,
                                                    rotations=rotations, cov3D_precomp=cov3D_precomp, shs_rest=shs_rest)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": vis.view(torch.bool),
            "radii": radii, "depth": depth, "tidx": tidx}


def render_motion(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None,
                  xyz_t=None, r_t=None, opacity=None):
    """Render externally supplied per-frame positions/rotations [REF gaussian_renderer/__init__.py:117-191]."""
    screenspace_points = _screenspace_points(pc)
    _wait_params(pc)
    rasterizer = GaussianRasterizer(raster_settings=_settings(viewpoint_camera, pc, bg_color, scaling_modifier))
    opacity = pc.get_opacity if opacity is None else opacity
    shs, colors_precomp = (pc.get_features, None) if override_color is None else (None, override_color)
    rendered_image, radii, depth, tidx = rasterizer(means3D=xyz_t, means2D=screenspace_points, shs=shs,
                                                    colors_precomp=colors_precomp, opacities=opacity,
                                                    scales=pc.get_scaling, rotations=r_t, cov3D_precomp=None)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii}


class SpeculativeRenderer:
    """render() for inference loops WITHOUT a host synchronisation per frame (extension; the eval harness's form of TrainStep's
    speculative binning).  The reference's rasterizer -- and render() in its exact mode -- reads the number of tile-splat instances back
    to the host in the middle of every frame [REF eval.py:208-215 times exactly that]; here the first frame is exact and reports R,
    later frames run in capacity mode with `margin` x the largest R seen and write {R, overflow} into a slot of a ring of status
    words; `flush()` (and a full ring) reads the slots back in ONE synchronisation, re-renders the frames that overflowed exactly and
    raises the high-water mark.  `__call__` returns the render dict; the image of a frame that later turns out to have overflowed is
    replaced IN PLACE by the exact re-render (same tensor object), so results collected before flush() are valid after it."""

    KEY_BITS = 24               # the depth-key window promised to capacity-mode frames (three sort passes instead of four)
    KEY_MIN_MARGIN = 1 << 19    # keys of room demanded on either side of the range seen so far

    def __init__(self, pc, pipe, bg_color, margin=1.25, slots=32, depth_key_speculation=True):
        self.pc, self.pipe, self.bg = pc, pipe, bg_color
        self.margin, self.slots = float(margin), int(slots)
        self.capacity = 0
        self.depth_key_speculation = bool(depth_key_speculation)
        self._key_lo, self._key_hi = None, None
        self._status = None
        self._pending = []          # (slot, camera, time, it, pkg)
        self.rerendered = 0

    def _exact(self, viewpoint_camera, time, it):
        """An exact-mode frame (one host synchronisation inside) that reports R and the visible depth keys' range."""
        st = self._status[self.slots]               # (the row behind the ring)
        st[4:6].copy_(self._key_none, non_blocking=True)
        pkg = render(viewpoint_camera, self.pc, self.pipe, self.bg, time=time, it=it, binning=(0, st[0:3], None, st[4:6]))
        h = st.cpu()
        r, lo, hi = int(h[0]), int(h[4]) & 0xFFFFFFFF, int(h[5]) & 0xFFFFFFFF
        self.capacity = max(self.capacity, 1024, int(self.margin * max(1, r)))
        if lo <= hi:
            self._key_lo = lo if self._key_lo is None else min(self._key_lo, lo)
            self._key_hi = hi if self._key_hi is None else max(self._key_hi, hi)
        return pkg

    def _promise(self):
        if not self.depth_key_speculation or self._key_lo is None:
            return None
        slack = (1 << self.KEY_BITS) - 1 - (self._key_hi - self._key_lo)
        if slack < 2 * self.KEY_MIN_MARGIN:
            return None
        return self.KEY_BITS, max(1, self._key_lo - slack // 2)

    def __call__(self, viewpoint_camera, time=None, it=1):
        dev = self.bg.device
        if self._status is None:
            self._status = torch.zeros(self.slots + 1, 8, dtype=torch.int32, device=dev)
            self._key_none = torch.tensor([-1, 0], dtype=torch.int32, device=dev)      # {0xFFFFFFFF, 0}: nothing reported
        if self.capacity <= 0:                      # exact frame: learns R and the key range (the call synchronises anyway)
            return self._exact(viewpoint_camera, time, it)
        if len(self._pending) >= self.slots:
            self.flush()
        slot = len(self._pending)
        pkg = render(viewpoint_camera, self.pc, self.pipe, self.bg, time=time, it=it,
                     binning=(self.capacity, self._status[slot][0:3], self._promise()))
        self._pending.append((slot, viewpoint_camera, time, it, pkg))
        return pkg

    def flush(self):
        """One synchronisation for all frames since the last flush; returns the number of frames that had to be rendered again (instance
        count beyond the capacity, or a visible depth key outside the promised window: both raise the frame's overflow word)."""
        if not self._pending:
            return 0
        st = self._status[:len(self._pending)].cpu()          # (waits for the frames)
        again = 0
        hi = 0
        pending, self._pending = self._pending, []
        for slot, cam, time, it, pkg in pending:
            hi = max(hi, int(st[slot, 0]))
            if int(st[slot, 1]) != 0:
                exact = self._exact(cam, time, it)
                for k, v in exact.items():
                    if torch.is_tensor(v) and torch.is_tensor(pkg.get(k)) and pkg[k].shape == v.shape:
                        pkg[k].copy_(v)
                again += 1
        self.capacity = max(self.capacity, int(self.margin * hi))
        self.rerendered += again
        return again
